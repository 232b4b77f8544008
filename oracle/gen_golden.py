"""TEST INFRASTRUCTURE (container-only): record golden traces from the reference itself.

Imports the reference's frozen legacy env (ref_harness.py), drives it with seeded action
streams and writes tests/golden/<name>.npz.  The fixtures are data only: domain coordinates,
actions, and the reference's outputs/state after every step.

usage: python oracle/gen_golden.py            (rewrites every fixture)
"""
from __future__ import annotations

import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_harness as H  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# (fixture name, domain, stream kind, seed, T)
TRACES = [
    ("boundary0_uniform_s0", "boundary0", "uniform", 0, 200),     # BASELINE.json configs[0]
    ("boundary0_biased_s1", "boundary0", "biased", 1, 700),
    ("boundary0_uniform_s7", "boundary0", "uniform", 7, 1500),    # long enough for 100-failure truncations
    ("boundary16_biased_s2", "boundary16", "biased", 2, 500),
    ("boundary15_uniform_s3", "boundary15", "uniform", 3, 200),
    ("boundary15_biased_s5", "boundary15", "biased", 5, 300),
    ("test1_biased_s42", "test1", "biased", 42, 300),
    ("dolphine3_biased_s0", "dolphine3", "biased", 0, 300),
    ("random1_1_biased_s1", "random1_1", "biased", 1, 300),
    ("basic_biased_s4", "basic", "biased", 4, 40),       # 5-vertex ring: every step ends the episode at B:141-143 (no extraction)
    ("star_biased_s6", "star", "biased", 6, 300),
]


def action_for_target(env, rule_value, target):
    """Invert B:115-123 / D:112-137 for the env's current point environment."""
    pe = env.current_point_environment
    p0, p1 = pe.neighbors[3], pe.neighbors[4]
    theta = math.atan2(p1.y - p0.y, p1.x - p0.x)
    X = (target[0] - p0.x) / pe.base_length
    Y = (target[1] - p0.y) / pe.base_length
    x = math.cos(theta) * X + math.sin(theta) * Y
    y = -math.sin(theta) * X + math.cos(theta) * Y
    return np.array([rule_value, x, y], np.float32)


def targeted_trace():
    """boundary0 with hand-placed candidate points: vertex-on-ray and on-edge point-in-polygon
    cases, a find_same_point hit, then a scripted sequence of rule -1 / +1 / 0 extractions."""
    pts = H.domain_points("boundary0")
    env = H.make_env(pts)
    env.reset()
    plan = [
        (0.0, (3.0, 2.0)),      # ray through vertices (10,2)
        (0.0, (3.0, 6.0)),      # on the top horizontal edge
        (0.0, (0.0, 3.0)),      # exactly a boundary vertex
        (0.0, (-1.0, 3.0)),     # outside, ray crosses the polygon
        (0.0, (6.0, 0.0)),      # ray through the apex (12,0)
        (0.0, (6.0, -6.0)),     # bottom vertex
        (0.0, (13.0, 0.0)),     # outside right, y of apex
        (0.0, (0.0004, 3.0)),   # find_same_point hit candidate
        (0.0, (1.0004, 0.9996)),
        (-1.0, (1.0, 1.0)),
        (1.0, (1.0, 1.0)),
        (0.0, (1.0, 1.0)),
        (0.0, (1.0, 2.0)),
        (0.0, (1.0, 3.0)),
        (0.0, (2.0, 1.0)),
        (-0.5, (1.0, 1.0)),     # thresholds are inclusive: <= -0.5 is rule -1
        (0.5, (1.0, 1.0)),      # >= 0.5 is rule +1
        (-0.49999, (2.0, 2.0)),
        (0.49999, (2.0, 3.0)),
    ]
    # then: greedy scripted meshing -- try the ideal "square corner" point for the current reference vertex
    actions = []

    class _Rec:
        pass

    rec = _Rec()
    rec.actions = actions
    for rule, tgt in plan:
        a = action_for_target(env, rule, tgt)
        actions.append(a)
        _, _, done, _ = env.step(a)
        if done:
            env.reset()
    rng = np.random.default_rng(123)
    for _ in range(400):
        pe = env.current_point_environment
        ref, right, left = pe.neighbors[3], pe.neighbors[4], pe.neighbors[2]
        # parallelogram completion point + small jitter on the 1e-4 grid
        tx = right.x + left.x - ref.x + float(rng.integers(-2000, 2001)) * 1e-4
        ty = right.y + left.y - ref.y + float(rng.integers(-2000, 2001)) * 1e-4
        rule = float(rng.choice([-1.0, 0.0, 0.0, 0.0, 1.0]))
        a = action_for_target(env, rule, (tx, ty))
        actions.append(a)
        _, _, done, _ = env.step(a)
        if done:
            env.reset()
    return pts, np.stack(actions).astype(np.float32)


def save(name, tr):
    path = os.path.join(OUT, name + ".npz")
    n0 = tr["domain_xy"].shape[0]
    small = np.int16 if n0 + len(tr["actions"]) < 32000 else np.int32
    tr = dict(tr)
    for k in ("ring_ids", "cand_ids", "reset_cand_ids"):
        tr[k] = tr[k].astype(small)
    np.savez_compressed(path, **tr)
    print(f"{name}: T={len(tr['actions'])} n0={n0} valid={int(tr['valid'].sum())} done={int(tr['done'].sum())} "
          f"complete={int((tr['done'] & tr['complete']).sum())} none={int(tr['obs_none'].sum())} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


def export_fixture():
    """The .inp text the reference writes for the first completed episode of the boundary0_uniform_s7 stream."""
    import contextlib, io, json, tempfile
    pts = H.domain_points("boundary0")
    acts = H.uniform_actions(7, 1500)
    env = H.make_env(pts)
    env.reset()
    for t in range(len(acts)):
        _, _, done, info = env.step(acts[t])
        if done and info["is_complete"]:
            with tempfile.NamedTemporaryFile("r", suffix=".inp") as f, contextlib.redirect_stdout(io.StringIO()):
                env.write_generated_elements_2_file(f.name)
                text = open(f.name).read()
            json.dump({"trace": "boundary0_uniform_s7", "step": t, "n_elements": len(env.generated_meshes), "inp": text},
                      open(os.path.join(OUT, "inp_boundary0_uniform_s7.json"), "w"))
            print(f"inp fixture: step {t}, {len(env.generated_meshes)} elements, {len(text)} bytes")
            return
        if done:
            env.reset()


def reference_element_record(mesh):
    """The eight per-element measures, each computed by the reference's own Mesh methods."""
    angles = [mesh.vertices[i].to_find_clockwise_angle(mesh.vertices[(i + 1) % 4], mesh.vertices[i - 1]) for i in range(4)]
    return [math.degrees(min(angles)), math.degrees(max(angles)), mesh.get_quality(type='s_jacobian'),
            mesh.get_quality(type='stretch'), mesh.get_quality(type='taper'), mesh.get_quality(type='robust'),
            mesh.compute_area()[0], mesh.get_quality()]


def quality_fixture():
    """Per-element quality records of every element the reference generates on three action streams, plus the
    write_2_file JSON (rl/boundary_env.py:648-669) of the first completed boundary0 episode."""
    import json, tempfile
    quads, recs, ep = [], [], []
    w2f = None
    for dom, kind, seed, T in (("boundary0", "biased", 1, 700), ("boundary16", "biased", 2, 500), ("random1_1", "biased", 1, 300)):
        pts = H.domain_points(dom)
        acts = (H.uniform_actions if kind == "uniform" else H.biased_actions)(seed, T)
        env = H.make_env(pts)
        env.reset()
        episode = 0
        for t in range(T):
            _, _, done, info = env.step(acts[t])
            if done or t == T - 1:
                for m in env.generated_meshes:
                    quads.append([[float(v.x), float(v.y)] for v in m.vertices])
                    recs.append(reference_element_record(m))
                    ep.append(len(ep_keys))
                ep_keys.append((dom, episode))
                if w2f is None and dom == "boundary0" and done and info["is_complete"]:
                    with tempfile.NamedTemporaryFile("r", suffix=".json") as f:
                        env.write_2_file(f.name)
                        w2f = {"trace": "boundary0_biased_s1", "step": t, "json": json.load(open(f.name))}
                episode += 1
                env.reset()
    np.savez_compressed(os.path.join(OUT, "quality_quads.npz"), quad_xy=np.asarray(quads, np.float64),
                        expected=np.asarray(recs, np.float64), episode=np.asarray(ep, np.int32))
    json.dump(w2f, open(os.path.join(OUT, "write2file_boundary0_biased_s1.json"), "w"))
    print(f"quality fixture: {len(quads)} elements in {len(ep_keys)} episodes; write_2_file at step {w2f['step']}: "
          f"{len(w2f['json']['nodes'])} nodes, {len(w2f['json']['elements'])} elements")


ep_keys = []


# move() API fixtures: (fixture name, domain or literal ring, seed, T, reset_on_done)
HEXAGON = [(0.0, 0.0), (0.0, 1.0), (0.0, 2.0), (1.0, 2.0), (1.0, 1.0), (1.0, 0.0)]
OCTAGON = [(0.0, 0.0), (0.0, 1.0), (0.0, 2.0), (0.0, 3.0), (1.0, 3.0), (1.0, 2.0), (1.0, 1.0), (1.0, 0.0)]
HEPTAGON = [(0.0, 0.0), (0.0, 1.0), (0.0, 2.0), (1.0, 2.5), (2.0, 2.0), (2.0, 1.0), (2.0, 0.0)]   # odd ring: ends on 5
MOVE_TRACES = [
    ("move_boundary0_s0", "boundary0", 0, 700, True),
    ("move_boundary16_s1", "boundary16", 1, 300, True),
    ("move_random1_1_s2", "random1_1", 2, 250, True),
    ("move_hexagon_s3", HEXAGON, 3, 120, False),     # completes in one or two moves; then the reference raises
    ("move_octagon_s4", OCTAGON, 4, 200, False),
    ("move_heptagon_s5", HEPTAGON, 5, 120, False),
    # more of move()'s smooth_pave branch (B:405-426), incl. the front smoother's side_vertex constructions (sharp corners)
    ("move_dolphine3_s11", "dolphine3", 11, 500, True),
    ("move_star_s13", "star", 13, 500, True),
    ("move_rand903_s903", "RANDOM903", 903, 500, True),
    ("move_rand928_s928", "RANDOM928", 928, 500, True),
    ("move_rand906_s906", "RANDOM906", 906, 500, True),
    # a self-touching generated polygon (doubled spike): coincident front vertices make the front smoother's vertex
    # constructions divide by zero -- with a NumPy-scalar operand the reference warns and goes on (moves 993-995: `warned`),
    # with Python operands it raises (move 996: code 4).  Inputs: the parity campaign's distribution.
    ("move_rand7023_c1", "RANDOM7023", 1, 1200, True, "campaign"),
]


# (fixture name, domain, seed, T, smooth after every k-th accepted element, iteration cap)
SMOOTH_TRACES = [
    ("smooth_boundary0_s1", "boundary0", 1, 600, 3, 400),
    ("smooth_boundary0_s8", "boundary0", 8, 500, 2, 5),      # the iteration cap ends the sweeps, not the 0.001 rule
    ("smooth_boundary16_s2", "boundary16", 2, 400, 4, 400),
    ("smooth_random1_1_s3", "random1_1", 3, 400, 3, 400),
]


def main_smooth():
    for name, dom, seed, T, every, iteration in SMOOTH_TRACES:
        tr = H.record_smooth_trace(H.domain_points(dom), H.biased_actions(seed, T), every, iteration=iteration, max_v=4096)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **tr)
        print(f"{name}: {T} steps, {int(tr['valid'].sum())} valid, {int(tr['n_calls'])} smooth_pave calls, sweeps "
              f"{tr['call_sweeps'].tolist() if tr['n_calls'] else []}")


# finished meshes: (fixture name, domain, stream kind, seed, T, iteration cap)
FINAL_SMOOTH_TRACES = [
    ("smoothfinal_boundary0_s7", "boundary0", "uniform", 7, 3000, 400),
    ("smoothfinal_boundary0_s21", "boundary0", "uniform", 21, 3000, 400),
    ("smoothfinal_boundary0_s33", "boundary0", "biased", 33, 2500, 12),    # the iteration cap ends the sweeps
    ("smoothfinal_star_s6", "star", "biased", 6, 1200, 400),
    ("smoothfinal_ring13_s4", "RING13", "biased", 4, 1500, 400),           # odd ring: episodes end on a front of 5
    ("smoothfinal_ring21_s9", "RING21", "biased", 9, 2500, 400),
]


def odd_ring(n, radius):
    """clockwise n-gon with a mild radial wobble, coordinates rounded to 2 decimals (data, not reference code)"""
    return [(round(radius * (1 + 0.08 * math.sin(3 * k)) * math.cos(-2 * math.pi * k / n), 2),
             round(radius * (1 + 0.08 * math.sin(3 * k)) * math.sin(-2 * math.pi * k / n), 2)) for k in range(n)]


def main_final_smooth():
    for name, dom, kind, seed, T, iteration in FINAL_SMOOTH_TRACES:
        acts = (H.uniform_actions if kind == "uniform" else H.biased_actions)(seed, T)
        pts = odd_ring(int(dom[4:]), 0.9 * int(dom[4:]) / 6) if dom.startswith("RING") else H.domain_points(dom)
        tr = H.record_final_smooth_trace(pts, acts, iteration=iteration, max_calls=24)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **tr)
        print(f"{name}: {T} steps, {int(tr['n_calls'])} smooth() calls, sweeps "
              f"{tr['call_sweeps'].tolist() if tr['n_calls'] else []}, front {tr['call_nr'].tolist() if tr['n_calls'] else []}, "
              f"branch visits {tr['call_branch'].sum(axis=0).tolist() if tr['n_calls'] else []}")


# front smoothing: (fixture name, domain, seed, T, smooth after every k-th accepted element)
FRONT_SMOOTH_TRACES = [
    ("smoothfront_boundary0_s1", "boundary0", 1, 700, 3),
    ("smoothfront_boundary0_s9", "boundary0", 9, 600, 2),
    ("smoothfront_boundary16_s2", "boundary16", 2, 400, 3),
    ("smoothfront_random1_1_s3", "random1_1", 3, 400, 2),
    ("smoothfront_star_s5", "star", 5, 400, 2),
]


def main_front_smooth():
    for name, dom, seed, T, every in FRONT_SMOOTH_TRACES:
        tr = H.record_front_smooth_trace(H.domain_points(dom), H.biased_actions(seed, T), every)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **tr)
        nc = int(tr["n_calls"])
        print(f"{name}: {T} steps, {int(tr['valid'].sum())} valid, {nc} smooth_pave(interior=False) calls, "
              f"{int(tr['call_raised'].sum()) if nc else 0} raised, {int(tr['call_obs_none'].sum()) if nc else 0} None")


def main_move(only=None):
    for name, dom, seed, T, reset_on_done, *kind in MOVE_TRACES:
        if only and name not in only:
            continue
        if isinstance(dom, str) and dom.startswith("RANDOM"):
            from reinforcementlearning4meshgeneration_amd.domains import random_domain
            pts = random_domain(int(dom[6:]))
        else:
            pts = H.domain_points(dom) if isinstance(dom, str) else dom
        p, ty = (H.campaign_move_inputs if kind == ["campaign"] else H.move_inputs)(seed, T)
        tr = H.record_move_trace(pts, p, ty, reset_on_done=reset_on_done)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **tr)
        print(f"{name}: {T} moves, {int(tr['valid'].sum())} valid, codes {np.bincount(tr['code'], minlength=4).tolist()}, "
              f"done {int(tr['done'].sum())}, complete {int(tr['complete'].sum())}, max not_valid {int(tr['n_not_valid'].max())}, "
              f"through smooth_pave {int(tr['smoothed'].sum())}, with NumPy zero-divisor warnings {int((tr['warned'] > 0).sum())}")


# (fixture name, domain, seed, T): accepted find_same_point extractions (B:165-175 -> M:623-629)
SAMEPOINT_TRACES = [
    ("boundary0_samepoint", "boundary0", 5, 260),
    ("star_samepoint", "star", 6, 220),
    ("random1_1_samepoint", "random1_1", 7, 220),
]


def samepoint_trace(dom, seed, T):
    """A rule-0 action whose decoded point lies within 0.001 of a ring vertex while the rule -1 quad [i-1, i, i+1, i+2] is
    acceptable: the reference reuses that quad with `rule = 0` kept (B:165-175), so update_boundary takes its
    "no new vertex" branch (M:623-629) and the reward bookkeeping is rule 0's.  The reference env cannot be copied or
    undone, so every probe replays the episode's action history on a fresh env.  Between such extractions the stream is
    the biased random one; near-vertex points that the reference rejects (rule -1 quad invalid, point outside) are kept
    in the stream too."""
    pts = H.domain_points(dom)

    def replay(hist):
        e = H.make_env(pts)
        e.reset()
        for a in hist:
            e.step(a)
        return e

    env = H.make_env(pts)
    env.reset()
    rng = np.random.default_rng(seed)
    hist, actions = [], []
    offsets = [(0, 0), (4, 3), (-4, 3), (4, -3), (-4, -3), (7, 0), (0, 7), (-7, 0), (0, -7), (9, 4), (-2, -9), (0.4, 0.3)]
    accepted = 0
    while len(actions) < T:
        a = None
        p = replay(hist)
        nb = len(p.generated_meshes)
        p.step(np.array([-1.0, 0.5, 0.5], np.float32))
        rule_m1_ok = len(p.generated_meshes) > nb
        ring = env.updated_boundary.vertices
        i, n = ring.index(env.current_point_environment.reference_point), len(ring)
        want = rng.random()
        if rule_m1_ok and want < 0.7:
            ks = [i + 2, i, i + 1, i - 1, i + int(rng.integers(3, max(4, n - 2)))]
            rng.shuffle(ks)
            for k in ks:
                v = ring[k % n]
                order = rng.permutation(len(offsets))
                for o in order:
                    dx, dy = offsets[o][0] * 1e-4, offsets[o][1] * 1e-4
                    c = action_for_target(env, float(rng.choice([0.0, 0.3, -0.45, 0.49])), (v.x + dx, v.y + dy))
                    if not (abs(c[1]) <= 1.5 and 0 <= c[2] <= 1.5):
                        continue
                    q = replay(hist)
                    qb, qv = len(q.generated_meshes), len(q.boundary.vertices)
                    q.step(c)
                    if len(q.generated_meshes) > qb and len(q.boundary.vertices) == qv:
                        a = c
                        accepted += 1
                        break
                if a is not None:
                    break
        elif want > 0.9:
            # a near-vertex point without the guarantee (rejected when the rule -1 quad is invalid or the point is outside)
            v = ring[int(rng.integers(n))]
            dx, dy = offsets[int(rng.integers(len(offsets)))]
            c = action_for_target(env, 0.0, (v.x + dx * 1e-4, v.y + dy * 1e-4))
            if abs(c[1]) <= 1.5 and 0 <= c[2] <= 1.5:
                a = c
        if a is None:
            a = H.biased_actions(int(rng.integers(1 << 30)), 1)[0]
        actions.append(a)
        hist.append(a)
        _, _, done, _ = env.step(a)
        if done:
            env.reset()
            hist = []
    return pts, np.stack(actions).astype(np.float32), accepted


def main_samepoint():
    for name, dom, seed, T in SAMEPOINT_TRACES:
        pts, acts, accepted = samepoint_trace(dom, seed, T)
        tr = H.record_trace(pts, acts)
        hits = int(((tr["valid"] == 1) & (np.abs(acts[:, 0]) < 0.5) & np.isnan(tr["new_xy"][:, 0])).sum())
        assert hits == accepted, (hits, accepted)
        save(name, tr)
        print(f"  {name}: {hits} rule-0 extractions without a new vertex (accepted find_same_point)")


# (fixture name, domain, seed, T, p_valid): streams steered towards accepted extractions (several whole episodes each)
GUIDED_TRACES = [
    ("half_wheel_guided_s4", "half_wheel", 4, 160, 0.8),
    ("basic1_guided_s9", "basic1", 9, 260, 0.8),
    ("boundary0_guided_s3", "boundary0", 3, 260, 0.85),
]


def guided_trace(dom, seed, T, p_valid):
    """With probability p_valid the next action is one the reference accepts at the current state -- found by trying
    rule -1, rule +1 and jittered parallelogram-completion points on a replay of the episode's history (the env cannot
    be copied) -- otherwise a biased random one.  Gives fixtures whose steps are mostly extractions."""
    pts = H.domain_points(dom)

    def accepted(hist, a):
        e = H.make_env(pts)
        e.reset()
        for h in hist:
            e.step(h)
        nb = len(e.generated_meshes)
        e.step(a)
        return len(e.generated_meshes) > nb

    env = H.make_env(pts)
    env.reset()
    rng = np.random.default_rng(seed)
    hist, actions = [], []
    while len(actions) < T:
        a = None
        if rng.random() < p_valid:
            pe = env.current_point_environment
            ref, right, left = pe.neighbors[3], pe.neighbors[4], pe.neighbors[2]
            cands = [np.array([float(rng.uniform(-1, -0.5)), float(rng.uniform(-1.5, 1.5)), float(rng.uniform(0, 1.5))], np.float32),
                     np.array([float(rng.uniform(0.5, 1)), float(rng.uniform(-1.5, 1.5)), float(rng.uniform(0, 1.5))], np.float32)]
            for _ in range(4):
                tx = right.x + left.x - ref.x + float(rng.integers(-3000, 3001)) * 1e-4 * pe.base_length
                ty = right.y + left.y - ref.y + float(rng.integers(-3000, 3001)) * 1e-4 * pe.base_length
                cands.append(action_for_target(env, float(rng.uniform(-0.49, 0.49)), (tx, ty)))
            for _ in range(10):     # the sub-box the biased stream draws from
                cands.append(np.array([float(rng.uniform(-0.49, 0.49)), float(rng.uniform(0.2, 1.0)), float(rng.uniform(0.3, 1.2))], np.float32))
            for k in rng.permutation(len(cands)):
                c = cands[k]
                if abs(c[1]) <= 1.5 and 0 <= c[2] <= 1.5 and accepted(hist, c):
                    a = c
                    break
        if a is None:
            a = H.biased_actions(int(rng.integers(1 << 30)), 1)[0]
        actions.append(a)
        hist.append(a)
        _, _, done, _ = env.step(a)
        if done:
            env.reset()
            hist = []
    return pts, np.stack(actions).astype(np.float32)


def main_guided():
    for name, dom, seed, T, p_valid in GUIDED_TRACES:
        pts, acts = guided_trace(dom, seed, T, p_valid)
        save(name, H.record_trace(pts, acts))


def plot_fixture():
    """PNG files the reference's save_meshes writes (general/mesh.py:1785-1792) for the first completed episode of the
    boundary0_biased_s1 stream, with the calls its callers make: the evaluation callback's (rl/baselines/CustomizeCallback.py:
    131-133: indexing=True, style='k-', dpi=30), testbed's (rl/baselines/testbed.py:189-191: quality=False, type=4, style='k-';
    dpi lowered to 40 to keep the fixture small) and a labelled one (quality=True, indexing=True, type=4).  Also
    get_quality(element, index) of every element for the indices that are functions of the quad (0, 1, 3, 4, 5).  The
    images depend on the matplotlib version (3.x Agg rasteriser): the fixture records it."""
    import tempfile
    import matplotlib
    pts = H.domain_points("boundary0")
    acts = H.biased_actions(1, 700)
    env = H.make_env(pts)
    env.reset()
    for t in range(len(acts)):
        _, _, done, info = env.step(acts[t])
        if done and info["is_complete"]:
            break
        if done:
            env.reset()
    table = {id(v): k for k, v in enumerate(env.boundary.vertices)}
    quads = np.array([[table[id(v)] for v in m.vertices] for m in env.generated_meshes], np.int32)
    vxy = np.array([[float(v.x), float(v.y)] for v in env.boundary.vertices], np.float64)
    out = dict(trace="boundary0_biased_s1", step=np.int32(t), quads=quads, vertex_xy=vxy, n0=np.int32(len(pts)),
               matplotlib_version=np.array(matplotlib.__version__),
               quality_index=np.array([0, 1, 3, 4, 5], np.int32),
               quality=np.array([[env.get_quality(m, k) for k in (0, 1, 3, 4, 5)] for m in env.generated_meshes], np.float64))
    calls = {"callback": dict(indexing=True, style='k-', dpi=30),
             "testbed": dict(quality=False, type=4, indexing=False, style='k-', dpi=40),
             "labelled": dict(quality=True, indexing=True, type=4, dpi=40)}
    for key, kw in calls.items():
        with tempfile.NamedTemporaryFile("rb", suffix=".png") as f:
            env.save_meshes(f.name, meshes=env.generated_meshes, **kw)
            out["png_" + key] = np.frombuffer(open(f.name, "rb").read(), np.uint8)
    np.savez_compressed(os.path.join(OUT, "plot_boundary0_biased_s1.npz"), **out)
    print(f"plot fixture: step {t}, {len(quads)} elements, PNG bytes {[len(out['png_' + k]) for k in calls]}")


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--smooth-only" in sys.argv:
        main_smooth()
        main_final_smooth()
        main_front_smooth()
        return
    if "--quality-only" in sys.argv:
        quality_fixture()
        plot_fixture()
        return
    if "--samepoint-only" in sys.argv:
        main_samepoint()
        return
    if "--guided-only" in sys.argv:
        main_guided()
        save("basic_biased_s4", H.record_trace(H.domain_points("basic"), H.biased_actions(4, 40)))
        return
    if "--move-only" in sys.argv:
        main_move(only=[a for a in sys.argv[1:] if a.startswith("move_")])
        return
    export_fixture()
    quality_fixture()
    plot_fixture()
    for name, dom, kind, seed, T in TRACES:
        pts = H.domain_points(dom)
        acts = (H.uniform_actions if kind == "uniform" else H.biased_actions)(seed, T)
        save(name, H.record_trace(pts, acts))
    pts, acts = targeted_trace()
    save("boundary0_targeted", H.record_trace(pts, acts))
    main_samepoint()
    main_guided()
    main_move()
    main_smooth()
    main_final_smooth()
    main_front_smooth()


if __name__ == "__main__":
    main()
