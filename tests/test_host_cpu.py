"""CPU (-m "not gpu"): host logic, domain constants against the reference's own numbers, the C-ABI library's
exports, and the no-CPU-fallback rule."""
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT, golden_names


def test_library_loads_and_exports_every_declared_symbol():
    from reinforcementlearning4meshgeneration_amd import _capi
    from reinforcementlearning4meshgeneration_amd.build import LIB_PATH
    assert os.path.exists(LIB_PATH), "run `python -m reinforcementlearning4meshgeneration_amd.build` first"
    L = _capi.load()
    header = open(os.path.join(ROOT, "include", "meshenv.h")).read()
    declared = sorted(set(re.findall(r"\b(meshenv_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/meshenv.h but not exported"
    assert sorted(_capi.EXPORTS) == declared
    assert L.meshenv_abi_version() == 1
    p = _capi.default_params()
    assert (p.neighbor_num, p.radius_num, p.fail_limit) == (6, 3, 100)
    assert p.radius == 4 and abs(p.max_ref_angle - 0.972 * np.pi) < 1e-15 and p.key_lambda == 0.618


def test_the_host_finds_and_validates_the_libms_atan2():
    """meshenv_atan2_exact (host-only, no GPU): the library locates glibc's 241 x 7 atan table in the libm image of this
    process and its restatement of __ieee754_atan2 (csrc/meshenv_libm.h) agrees with math.atan2's libm on the 2^18
    validation arguments -- on this image's glibc 2.35 it must, or the device's tie-breaker for quantised angles is the
    weaker correctly-rounded stand-in."""
    from reinforcementlearning4meshgeneration_amd import _capi
    assert _capi.load().meshenv_atan2_exact() == 1


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the product path must fail loudly, never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from reinforcementlearning4meshgeneration_amd import _capi, boundary
    from reinforcementlearning4meshgeneration_amd.vec_env import MeshVecEnv
    with pytest.raises(_capi.MeshEnvError):
        MeshVecEnv([boundary(0)], n_envs=4)
    import ctypes as C
    L = _capi.load()
    h = C.c_void_p()
    offs = (C.c_int32 * 2)(0, 4)
    xy = (C.c_double * 8)(0, 0, 0, 1, 1, 1, 1, 0)
    consts = (C.c_double * 3)(1, 1, 1)
    dom = (C.c_int32 * 1)(0)
    rc = L.meshenv_create(0, 1, offs, xy, consts, 1, dom, None, None, C.byref(h))
    assert rc == -2 and not h.value
    assert b"no HIP device" in L.meshenv_last_error(None) or b"hip" in L.meshenv_last_error(None).lower()


def test_product_code_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing in the package may import, load or link it."""
    pkg = os.path.join(ROOT, "reinforcementlearning4meshgeneration_amd")
    banned = re.compile(r"(from\s+oracle|import\s+oracle|libmeshenv_ref|libmeshenv_cpu|meshenv_ref_|ref_lib|ref_harness)")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                text = open(os.path.join(dirpath, f)).read()
                assert not banned.search(text), f"{f} references the oracle"


@pytest.mark.parametrize("name", golden_names())
def test_domain_constants_match_reference(name):
    from reinforcementlearning4meshgeneration_amd.domains import domain_constants
    tr = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    pts = [tuple(p) for p in tr["domain_xy"]]
    c = domain_constants(pts)
    ref = tr["consts"]
    # poly_area goes through BLAS dot (summation order may differ by CPU): 1e-13; the rest is exact
    assert abs(c.original_area - ref[0]) <= 1e-13 * abs(ref[0])
    assert c.average_edge_length == ref[1]
    assert c.est_min_l == ref[2] and c.est_crit_l == ref[3]


def test_builtin_and_random_domains():
    from reinforcementlearning4meshgeneration_amd import domains as D
    assert len(D.boundary(0)) == 30 and len(D.boundary(1)) == 44 and len(D.boundary(2)) == 40 and len(D.boundary(-1)) == 24
    for idx in (0, 1, 2, -1):
        assert D.signed_area2(D.boundary(idx)) < 0          # clockwise, like every domain the reference uses
    with pytest.raises(ValueError):
        D.boundary(7)
    for seed in range(20):
        pts = D.random_domain(seed)
        assert len(pts) >= 8 and D.signed_area2(pts) < 0
        assert all(round(x, 4) == x and round(y, 4) == y for x, y in pts)
        assert pts == D.random_domain(seed)                # deterministic
        assert all(pts[i] != pts[i - 1] for i in range(len(pts)))


def test_read_polygon_format(tmp_path):
    from reinforcementlearning4meshgeneration_amd.domains import read_polygon
    f = tmp_path / "d.json"
    f.write_text("[[100, 200], [150.5, 200], [150, 100]]\n[[0,0]]\n")     # only the first line counts
    assert read_polygon(f) == [(1.0, 2.0), (1.505, 2.0), (1.5, 1.0)]


def test_spaces_and_sharding():
    from reinforcementlearning4meshgeneration_amd import sharding
    from reinforcementlearning4meshgeneration_amd.vec_env import ACTION_HIGH, ACTION_LOW, make_spaces
    obs_space, act_space = make_spaces()
    assert obs_space.shape == (18,) and act_space.shape == (3,)
    assert np.allclose(act_space.low, [-1, -1.5, 0]) and np.allclose(act_space.high, [1, 1.5, 1.5])
    assert float(obs_space.low.min()) == -999 and float(obs_space.high.max()) == 999
    a = act_space.sample()
    assert a.dtype == np.float32 and np.all(a >= ACTION_LOW) and np.all(a <= ACTION_HIGH)
    for n, w in ((4096, 8), (10, 3), (7, 8), (32768, 8)):
        blocks = [sharding.shard_range(n, w, r) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in blocks]
        assert max(sizes) - min(sizes) <= 1


def test_helper_placement_table_of_the_group_kernel():
    """csrc/meshenv_kernels.h deals the m pending updates of a workgroup and their m reward helpers over the waves of
    four SIMDs with a nibble table; this mirrors the formulas and checks, for every group size and m, that each pending
    env gets exactly one update wave and exactly one helper, on distinct waves, within the waves a SIMD has."""
    import re
    src = open(os.path.join(ROOT, "reinforcementlearning4meshgeneration_amd", "csrc", "meshenv_kernels.h")).read()
    t16, t8 = re.search(r"t_lo = G >= 16 \? (0x[0-9a-f]+)ULL : (0x[0-9a-f]+)ULL", src).groups()
    t_hi = int(re.search(r"t_hi = (0x[0-9a-f]+)ULL", src).group(1), 16)
    for G in (4, 8, 16):
        t_lo = int(t16 if G >= 16 else t8, 16)
        per_simd = G // 4
        for m in range(1, G // 2 + 1):
            hcw = ((t_lo if m <= 4 else t_hi) >> (16 * ((m - 1) & 3))) & 0xffff
            mains, helpers = [], []
            for s in range(4):
                for r in range(per_simd):
                    j = 4 * r + s
                    if j < m:
                        mains.append(j)
                        continue
                    mains_here = (m + 3 - s) >> 2
                    t = r - mains_here
                    here = (hcw >> (4 * s)) & 15
                    below = sum((hcw >> (4 * q)) & 15 for q in range(s))
                    if 0 <= t < here:
                        helpers.append(below + t)
            assert sorted(mains) == list(range(m)), (G, m, mains)
            assert sorted(helpers) == list(range(m)), (G, m, helpers)


def test_every_hip_entry_point_runs_on_the_handles_device_and_restores_the_callers():
    """A process may hold handles on several GPUs: each C-ABI function that launches, copies or synchronises must switch
    to the handle's device through the save/restore guard, never through a bare hipSetDevice (which would leave the
    caller's -- torch's -- current device changed).  Checked on the source: a one-GPU box cannot show a wrong device."""
    src = open(os.path.join(ROOT, "reinforcementlearning4meshgeneration_amd", "csrc", "meshenv_hip.hip")).read()
    body = src[src.index('extern "C" {'):]
    bare = [m.start() for m in re.finditer(r"hipSetDevice\(", body)]
    assert not bare, "bare hipSetDevice in an entry point: use DeviceGuard / MESHENV_ON_DEVICE"
    # split into top-level function bodies and check those that touch the runtime
    funcs = re.split(r"\n(?=(?:static\s+)?(?:int|void|const char \*)\s+\*?[a-z_]+\()", body)
    needs = re.compile(r"hipLaunchKernelGGL|hipMemcpy|hipStreamSynchronize|hipEventRecord|hipEventCreate|hipMalloc\(|hipFree\(|hipDeviceSynchronize")
    checked = 0
    for f in funcs:
        name = re.match(r"(?:static\s+)?(?:int|void|const char \*)\s+\*?([a-z_0-9]+)\(", f)
        if not name or not needs.search(f) or "MESHENV_STAMPS" in f.split("{", 1)[0]:
            continue
        if name.group(1).startswith("meshenv_debug_"):
            continue
        assert re.search(r"MESHENV_ON_DEVICE\(|DeviceGuard\s+guard\(", f), f"{name.group(1)} has no device guard"
        checked += 1
    assert checked >= 12


def test_default_log_capacity_follows_the_smoothers_lds_formulas():
    """BoudaryEnv sizes its element log so that smooth() / smooth_pave() stay usable (vec_env.smoothing_log_capacity); the
    byte counts it assumes are the ones csrc/meshenv_smooth.h computes."""
    from reinforcementlearning4meshgeneration_amd.vec_env import smoothing_log_capacity
    src = open(os.path.join(ROOT, "reinforcementlearning4meshgeneration_amd", "csrc", "meshenv_smooth.h")).read()
    assert "constexpr int kSmoothMaxDeg = 16;" in src
    assert "return (size_t)(ring_cap + log_cap) * sizeof(double2) + (size_t)log_cap * (8 + kSmoothMaxDeg * 2 + 2 + 2) + 64;" in src
    assert "return V * sizeof(double2) + V * kSmoothMaxDeg * 2 + (size_t)ring_cap * 2 + 2 * kSmoothMaxDeg * 2 + V + 64;" in src
    assert "return V * sizeof(double2) + V * kSmoothMaxDeg * 2 + (size_t)ring_cap * 2 + V * 3 + 64;" in src

    def interior(c, l): return (c + l) * 16 + l * (8 + 32 + 2 + 2) + 64
    def front(c, l): return (c + l) * 16 + (c + l) * 32 + c * 2 + 64 + (c + l) + 64
    def final(c, l): return (c + l) * 16 + (c + l) * 32 + c * 2 + (c + l) * 3 + 64
    for ring in (6, 30, 120, 196, 272, 640, 1500):
        cap = (ring + 15) // 16 * 16
        lc = smoothing_log_capacity(ring)
        assert max(interior(cap, lc), front(cap, lc), final(cap, lc)) <= 160 * 1024 and cap + lc <= 65535
        if lc < 4096:   # tight: one more logged vertex would not fit
            assert max(interior(cap, lc + 1), front(cap, lc + 1), final(cap, lc + 1)) > 160 * 1024
    assert smoothing_log_capacity(30) >= 2400 and smoothing_log_capacity(272) >= 2000
