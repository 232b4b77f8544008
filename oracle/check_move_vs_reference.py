"""TEST INFRASTRUCTURE (container-only): drive the reference's move() and the C oracle's meshenv_ref_move side by side.

usage: python oracle/check_move_vs_reference.py [--domains 7023 7001 ...] [--moves N] [--seed S]
A domain is the seed of domains.random_domain (config-5 style generated polygon) or the name of a shipped domain.
Inputs are the parity campaign's (tools/parity_campaign.py): radius fraction U(0.05, 0.45), angle U(0.2, 1.5), type U(0, 1)
as Python floats.  Every return value is compared bit for bit; NumPy's RuntimeWarnings inside the reference (the
zero-divisor continuation of the front smoother's vertex constructions, oracle/meshenv_ref.c "the front smoother") are
counted.  Exits non-zero on a mismatch.
"""
from __future__ import annotations

import argparse
import contextlib
import io
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as H  # noqa: E402
from oracle.ref_lib import RefEnv  # noqa: E402
from reinforcementlearning4meshgeneration_amd.domains import random_domain  # noqa: E402


def drive(points, T, seed, label="", verbose=True):
    env = H.make_env(points)
    ref = RefEnv.from_points(points, cap_new=512)
    rng = np.random.default_rng(seed)
    assert np.array_equal(env.reset(static=True), ref.reset(static=True)[0])
    st = dict(moves=0, warned=0, raised=0, smoothed=0, mismatches=0, codes=[0, 0, 0, 0, 0])
    smoothed = [0]
    orig = env.smooth_pave

    def counting(*a, **k):
        smoothed[0] = 1
        return orig(*a, **k)
    env.smooth_pave = counting
    for t in range(T):
        p = [float(rng.uniform(0.05, 0.45)), float(rng.uniform(0.2, 1.5))]
        ty = float(rng.uniform(0, 1))
        smoothed[0] = 0
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    obs, rew, done, info = env.move(p, ty)
                code, comp = (1 if obs is None else 0), info["is_complete"]
            except UnboundLocalError:
                obs, done, comp, code = None, False, False, 2
            except (ValueError, ZeroDivisionError):
                obs, done, comp, code = None, True, False, 4
                st["raised"] += 1
        st["warned"] += bool(w)
        st["smoothed"] += smoothed[0]
        o_r, d_r, c_r, code_r = ref.move(np.array(p), ty)
        st["codes"][code] += 1
        mis = []
        if code != code_r:
            mis.append(f"code {code} vs {code_r}")
        elif code != 2 and (bool(done) != bool(d_r) or bool(comp) != bool(c_r)):
            mis.append(f"flags {done}/{comp} vs {d_r}/{c_r}")
        if code == 0 and code_r == 0 and not np.array_equal(np.asarray(obs, np.float32), o_r):
            mis.append("obs")
        ids_ref = [id(v) for v in env.updated_boundary.vertices]
        table = {id(v): k for k, v in enumerate(env.boundary.vertices)}
        rids, rxy = ref.ring()
        if code == code_r and code != 4:
            if [table[i] for i in ids_ref] != list(rids):
                mis.append("ring ids")
            elif not np.array_equal(np.array([(v.x, v.y) for v in env.updated_boundary.vertices], np.float64), rxy):
                mis.append("ring xy")
            if len(env.not_valid_points) != ref.not_valid_count():
                mis.append("not_valid")
        if mis:
            st["mismatches"] += 1
            if verbose:
                print(f"{label} move {t}: {mis} (warnings {len(w)}, through smooth_pave {smoothed[0]})", flush=True)
            if st["mismatches"] > 3:
                break
        st["moves"] += 1
        if done or code >= 2 or d_r or code_r >= 2:
            env.reset(static=True)
            ref.reset(static=True)
    return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--domains", nargs="*", default=["7023"])
    ap.add_argument("--moves", type=int, default=3000)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    bad = 0
    for d in args.domains:
        pts = random_domain(int(d)) if d.isdigit() else H.domain_points(d)
        st = drive(pts, args.moves, args.seed, label=d)
        print(d, st, flush=True)
        bad += st["mismatches"]
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
