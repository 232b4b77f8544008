"""TEST INFRASTRUCTURE (container-only): records MeshGeneration.extract_samples_2 of the reference on meshes it generated
itself (tests/golden/samples_*.npz: the mesh + the function's three return lists).  usage: python oracle/gen_samples_golden.py"""
from __future__ import annotations

import contextlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_harness as H  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
# (fixture, domain, seed, steps, (n_neighbor, n_radius, radius, index, quality_threshold)): the two parameter sets of the callers
CASES = [
    ("samples_boundary0_ebrd", "boundary0", 1, 260, (2, 3, 4, 1, 0.7)),          # general/EBRD.py:414
    ("samples_boundary0_post", "boundary0", 9, 200, (3, 3, 6, 5, 0.7)),          # general/post_processing.py:532
    ("samples_star_ebrd", "star", 5, 120, (2, 3, 4, 1, 0.5)),
]


def main():
    for name, dom, seed, T, (nn, nr, rad, idx, thr) in CASES:
        pts = H.domain_points(dom)
        env = H.make_env(pts)
        env.reset()
        acts = H.biased_actions(seed, T)
        used = T
        for t in range(T):
            _, _, done, _ = env.step(acts[t])
            if done:
                used = t + 1
                break
        table = {id(v): k for k, v in enumerate(env.boundary.vertices)}
        quads = np.array([[table[id(v)] for v in m.vertices] for m in env.generated_meshes], np.int32)
        vxy = np.array([(v.x, v.y) for v in env.boundary.vertices], np.float64)
        with contextlib.redirect_stdout(io.StringIO()):
            samples, types, outputs = env.extract_samples_2(env.generated_meshes, nn, nr, radius=rad, index=idx,
                                                            quality_threshold=thr)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), quads=quads, vertex_xy=vxy, n0=np.int32(len(pts)), actions=acts[:used],
                            params=np.array([nn, nr, rad, idx, thr], np.float64),
                            samples=np.array(samples, np.float64).reshape(len(samples), -1),
                            types=np.array(types, np.float64).reshape(-1), outputs=np.array(outputs, np.float64).reshape(-1, 2))
        print(f"{name}: {len(quads)} elements, {len(vxy)} vertices -> {len(samples)} samples of {len(samples[0]) if samples else 0} values, "
              f"types {np.unique(np.array(types).reshape(-1), return_counts=True)}")


if __name__ == "__main__":
    main()
