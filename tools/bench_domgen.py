"""Dev tool / profile driver: the domain pipeline on the device -- meshenv_create_random (uniform split) and
meshenv_create_random_density (the reference's calculate_density) for n rings.  Prints one JSON line with the algorithmic
bytes of the generator kernels (output only: 16 B per ring vertex; the input is a seed).  usage: bench_domgen.py [n]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reinforcementlearning4meshgeneration_amd import MeshVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
w = MeshVecEnv.from_random(256, 1); w.close()
torch.cuda.synchronize(); t0 = time.perf_counter()
env = MeshVecEnv.from_random(n, 1000)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
total = int(env._L.meshenv_max_ring(env._handle))
verts = sum(len(env.get_domain(k)[0]) for k in range(0, n, n // 256)) / 256 * n
env.close()
t0 = time.perf_counter()
envd = MeshVecEnv.from_random_density(n // 8, 5000, base_length=20.0)
torch.cuda.synchronize(); dtd = time.perf_counter() - t0
vd = sum(len(envd.get_domain(k)[0]) for k in range(0, n // 8, max(1, n // 8 // 256))) / 256 * (n // 8)
envd.close()
print(json.dumps({"profile_kernels": [
    {"match": "k_gen_rings(", "algorithmic_bytes_per_launch": 16.0 * verts, "note": f"{n} rings, {verts / n:.1f} vertices each on average: 16 B per ring vertex written"},
    {"match": "k_gen_count(", "algorithmic_bytes_per_launch": 4.0 * n, "note": "4 B per ring (its length)"},
    {"match": "k_dom_consts(", "algorithmic_bytes_per_launch": 16.0 * verts + 64.0 * n, "note": "ring read once + 64 B of constants per domain"},
    {"match": "k_gen_rings_density(", "algorithmic_bytes_per_launch": 16.0 * vd, "note": f"{n // 8} rings by calculate_density (base_length 20 px), {vd / (n // 8):.1f} vertices each"}],
    "create_random_s": dt, "create_random_density_s_incl_seed_probe": dtd}))
