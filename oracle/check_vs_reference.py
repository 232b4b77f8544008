"""TEST INFRASTRUCTURE (container-only): drive the reference env and the C oracle side by side.

usage: python oracle/check_vs_reference.py [domain ...] [--steps N] [--seed S] [--biased]
Exits non-zero on the first mismatch; everything is compared bit-for-bit.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as H  # noqa: E402
from oracle.ref_lib import RefEnv  # noqa: E402
from reinforcementlearning4meshgeneration_amd.domains import domain_constants  # noqa: E402


def compare_trace(points, tr, label="", verbose=True):
    """Replay a recorded reference trace through the C oracle; returns number of mismatching steps."""
    c = tr["consts"]
    env = RefEnv(np.array(points, np.float64), c[0], c[2], c[3])
    obs, none = env.reset()
    bad = 0
    if not np.array_equal(obs, tr["reset_obs"]):
        print(label, "reset obs mismatch", obs, tr["reset_obs"])
        bad += 1
    ids, keys = env.candidates()
    if not (np.array_equal(ids, tr["reset_cand_ids"]) and np.array_equal(keys, tr["reset_cand_keys"])):
        print(label, "reset candidate list mismatch")
        bad += 1
    T = len(tr["actions"])
    for t in range(T):
        obs, rew, done, comp, none = env.step(tr["actions"][t])
        msgs = []
        if none != bool(tr["obs_none"][t]):
            msgs.append(f"obs_none {none} vs {tr['obs_none'][t]}")
        elif not none and not np.array_equal(obs, tr["obs"][t]):
            msgs.append(f"obs\n  {obs}\n  {tr['obs'][t]}")
        if rew != tr["reward"][t]:
            msgs.append(f"reward {rew!r} vs {tr['reward'][t]!r}")
        if done != bool(tr["done"][t]) or comp != bool(tr["complete"][t]):
            msgs.append(f"done/complete {done}/{comp} vs {tr['done'][t]}/{tr['complete'][t]}")
        rids, _ = env.ring()
        n = tr["ring_len"][t]
        if len(rids) != n or not np.array_equal(rids, tr["ring_ids"][t, :n]):
            msgs.append(f"ring {rids} vs {tr['ring_ids'][t, :n]}")
        if env.ref_id() != tr["ref_id"][t]:
            msgs.append(f"ref {env.ref_id()} vs {tr['ref_id'][t]}")
        sc = env.scalars()
        if sc["n_elem"] != tr["n_elem"][t] or sc["failed_num"] != tr["failed_num"][t]:
            msgs.append(f"counters {sc} vs {tr['n_elem'][t]} {tr['failed_num'][t]}")
        if sc["current_area"] != tr["current_area"][t]:
            msgs.append(f"area {sc['current_area']!r} vs {tr['current_area'][t]!r}")
        ids, keys = env.candidates()
        m = min(tr["n_cand"][t], tr["cand_ids"].shape[1])
        if len(ids) != tr["n_cand"][t] or not np.array_equal(ids[:m], tr["cand_ids"][t, :m]) or \
                not np.array_equal(keys[:m], tr["cand_keys"][t, :m]):
            msgs.append("candidate list")
        if not np.isnan(tr["new_xy"][t, 0]):
            _, vxy = env.elements()
            if not np.array_equal(vxy[sc["n_vert"] - 1], tr["new_xy"][t]):
                msgs.append(f"new vertex {vxy[sc['n_vert'] - 1]} vs {tr['new_xy'][t]}")
        if msgs:
            bad += 1
            if verbose:
                print(f"{label} step {t} action {tr['actions'][t]}: " + "; ".join(msgs))
            if bad > 5:
                break
        if done and tr["auto_reset"]:
            env.reset()
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("domains", nargs="*", default=["boundary0"])
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--biased", action="store_true")
    args = ap.parse_args()
    total_bad = 0
    for name in args.domains:
        pts = H.domain_points(name)
        acts = (H.biased_actions if args.biased else H.uniform_actions)(args.seed, args.steps)
        try:
            tr = H.record_trace(pts, acts)
        except Exception as exc:  # some shipped domains crash the reference itself (zero-length edges)
            print(f"{name}: reference raised {type(exc).__name__}: {exc} -- skipped")
            continue
        dc = domain_constants(pts)
        ref_c = tr["consts"]
        mine = np.array([dc.original_area, dc.average_edge_length, dc.est_min_l, dc.est_crit_l])
        if not np.array_equal(mine, ref_c):
            print(name, "domain constants differ", mine, ref_c)
            total_bad += 1
        bad = compare_trace(pts, tr, label=name)
        print(f"{name}: {args.steps} steps, valid={int(tr['valid'].sum())}, done={int(tr['done'].sum())}, "
              f"mismatching steps={bad}")
        total_bad += bad
    sys.exit(1 if total_bad else 0)


if __name__ == "__main__":
    main()
