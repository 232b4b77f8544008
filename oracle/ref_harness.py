"""TEST INFRASTRUCTURE (container-only): import shim for the reference's frozen legacy env.

This module exists only to (1) validate oracle/meshenv_ref.c against the real
reference and (2) generate the golden fixtures under tests/golden/.  It reads
/root/reference at run time, so it is never used on the GPU box and never by
the product path.  Nothing here is shipped behaviour.

The stubbing mirrors the reference's own technique for running the env without
the RL stack (/root/reference/v2/tests/mesh_rl/test_boundary_env_equiv.py:17-151):
`gym` and `stable_baselines3.common.env_checker` are replaced by empty module
objects, and the `mesh_rl` / `mesh_rl.legacy` package __init__ files (which pull
gymnasium + SB3) are skipped by pre-registering bare packages with __path__ set.
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("MESHENV_REFERENCE_ROOT", "/root/reference")
_SRC = os.path.join(REFERENCE_ROOT, "v2", "src")

# action box of rl/boundary_env.py:27
ACTION_LOW = np.array([-1.0, -1.5, 0.0])
ACTION_HIGH = np.array([1.0, 1.5, 1.5])

# the 30 literal vertices of general/polygon.py:79-83 (boundary(index=0)); data, not code
BOUNDARY0 = [(0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (0, 6),
             (1, 6), (2, 6), (3, 6), (4, 6), (5, 6), (6, 6),
             (7, 5), (8, 4), (9, 3), (10, 2), (11, 1), (12, 0),
             (11, -1), (10, -2), (9, -3), (8, -4), (7, -5), (6, -6),
             (5, -5), (4, -4), (3, -3), (2, -2), (1, -1), (0, 0)]


def reference_available() -> bool:
    return os.path.isdir(os.path.join(_SRC, "mesh_rl", "legacy"))


_loaded = None


def load_reference():
    """Import the legacy env; returns (BoudaryEnvLegacy, Vertex, Segment, Boundary2D)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    os.environ.setdefault("MPLBACKEND", "Agg")
    if _SRC not in sys.path:
        sys.path.insert(0, _SRC)
    for name, sub in (("mesh_rl", "mesh_rl"), ("mesh_rl.legacy", os.path.join("mesh_rl", "legacy"))):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(_SRC, sub)]
            sys.modules[name] = m
    if "gym" not in sys.modules:
        gym = types.ModuleType("gym")
        spaces = types.ModuleType("gym.spaces")

        class _Box:
            def __init__(self, low, high, shape=None, dtype=None):
                if shape is not None:
                    shape = tuple(shape)
                    self.low = np.full(shape, low, dtype=np.float32)
                    self.high = np.full(shape, high, dtype=np.float32)
                else:
                    self.low = np.array(low, dtype=np.float32)
                    self.high = np.array(high, dtype=np.float32)
                    shape = self.low.shape
                self.shape = shape
                self.dtype = dtype or np.float32

        gym.Env = type("Env", (), {})
        spaces.Box = _Box
        gym.spaces = spaces
        sys.modules.update({"gym": gym, "gym.spaces": spaces})
    if "stable_baselines3" not in sys.modules:
        sb3 = types.ModuleType("stable_baselines3")
        common = types.ModuleType("stable_baselines3.common")
        ec = types.ModuleType("stable_baselines3.common.env_checker")
        ec.check_env = lambda *a, **k: None
        sys.modules.update({"stable_baselines3": sb3, "stable_baselines3.common": common,
                            "stable_baselines3.common.env_checker": ec})
    from mesh_rl.legacy.boundary_env_legacy import BoudaryEnvLegacy  # noqa: E402
    from mesh_rl.legacy.components_legacy import Vertex, Segment, Boundary2D  # noqa: E402
    _loaded = (BoudaryEnvLegacy, Vertex, Segment, Boundary2D)
    return _loaded


def domain_points(name: str):
    """Python-typed coordinates exactly as the reference builds them.

    'boundary0' -> int literals (general/polygon.py:79-83);
    anything else -> ui/domains/<name>.json, first line, each coord / 100
    (general/polygon.py:110-117).
    """
    if name == "boundary0":
        return [(x, y) for x, y in BOUNDARY0]
    path = os.path.join(REFERENCE_ROOT, "ui", "domains", name + ".json")
    with open(path, "r") as fr:
        verts = json.loads(fr.readline())
    return [(p[0] / 100, p[1] / 100) for p in verts]


def make_env(points):
    Env, Vertex, Segment, Boundary2D = load_reference()
    vs = [Vertex(x, y) for x, y in points]
    for i in range(len(vs)):           # as connect_vertices, general/mesh.py:1926-1930
        s = Segment(vs[i - 1], vs[i])
        vs[i - 1].assign_segment(s)
        vs[i].assign_segment(s)
    return Env(Boundary2D(vs))


def uniform_actions(seed: int, T: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return rng.uniform(ACTION_LOW, ACTION_HIGH, size=(T, 3)).astype(np.float32)


def biased_actions(seed: int, T: int) -> np.ndarray:
    """Half the actions are drawn from a sub-box that makes valid extractions likely
    (a[1] in [0.2,1], a[2] in [0.3,1.2]); rule type spread over all three rules."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(ACTION_LOW, ACTION_HIGH, size=(T, 3))
    pick = rng.random(T) < 0.5
    b = np.stack([rng.uniform(-1, 1, T), rng.uniform(0.2, 1.0, T), rng.uniform(0.3, 1.2, T)], axis=1)
    a[pick] = b[pick]
    return a.astype(np.float32)


def record_trace(points, actions: np.ndarray, auto_reset: bool = True) -> dict:
    """Drive the reference env with `actions` and record everything parity needs."""
    env = make_env(points)
    n0 = len(points)
    T = len(actions)
    reset_obs = env.reset()
    consts = dict(
        original_area=float(env.original_area),
        average_edge_length=float(env.average_edge_length),
        est_min_l=float(env.estimated_area_range[0]),
        est_crit_l=float(env.estimated_area_range[1]),
    )

    def ids_of(vlist):
        table = {id(v): k for k, v in enumerate(env.boundary.vertices)}
        return [table[id(v)] for v in vlist]

    def snapshot_cands():
        cv = env.candidate_vertices
        return ids_of([c[0] for c in cv]), [float(c[1]) for c in cv]

    out = dict(
        obs=np.zeros((T, 18), np.float32), obs_none=np.zeros(T, np.uint8),
        reward=np.zeros(T, np.float64), done=np.zeros(T, np.uint8), complete=np.zeros(T, np.uint8),
        ring_len=np.zeros(T, np.int32), ring_ids=np.full((T, n0), -1, np.int32),
        ref_id=np.full(T, -1, np.int32), n_elem=np.zeros(T, np.int32), failed_num=np.zeros(T, np.int32),
        current_area=np.zeros(T, np.float64), new_xy=np.full((T, 2), np.nan, np.float64),
        n_cand=np.zeros(T, np.int32), cand_ids=np.full((T, n0), -1, np.int32),
        cand_keys=np.full((T, n0), np.nan, np.float64), valid=np.zeros(T, np.uint8),
    )
    r_ids, r_keys = snapshot_cands()
    out["reset_cand_ids"] = np.array(r_ids, np.int32)
    out["reset_cand_keys"] = np.array(r_keys, np.float64)
    out["reset_ref_id"] = np.int32(ids_of([env.current_point_environment.reference_point])[0])
    for t in range(T):
        nverts_before = len(env.boundary.vertices)
        nelem_before = len(env.generated_meshes)
        obs, rew, done, info = env.step(actions[t])
        if obs is None:
            out["obs_none"][t] = 1
        else:
            out["obs"][t] = obs
        out["reward"][t] = rew
        out["done"][t] = done
        out["complete"][t] = info["is_complete"]
        ring = ids_of(env.updated_boundary.vertices)
        out["ring_len"][t] = len(ring)
        out["ring_ids"][t, :len(ring)] = ring
        out["ref_id"][t] = ids_of([env.current_point_environment.reference_point])[0]
        out["n_elem"][t] = len(env.generated_meshes)
        out["failed_num"][t] = env.failed_num
        out["current_area"][t] = env.current_area
        out["valid"][t] = len(env.generated_meshes) > nelem_before
        if len(env.boundary.vertices) > nverts_before:
            v = env.boundary.vertices[-1]
            out["new_xy"][t] = (v.x, v.y)
        cids, ckeys = snapshot_cands()
        out["n_cand"][t] = len(cids)
        # duplicates can push the list past n0 on tiny rings; clip for storage
        m = min(len(cids), n0)
        out["cand_ids"][t, :m] = cids[:m]
        out["cand_keys"][t, :m] = ckeys[:m]
        if done and auto_reset:
            env.reset()
    out.update(
        domain_xy=np.array(points, np.float64), actions=actions, reset_obs=reset_obs.astype(np.float32),
        consts=np.array([consts["original_area"], consts["average_edge_length"],
                         consts["est_min_l"], consts["est_crit_l"]], np.float64),
        auto_reset=np.uint8(auto_reset),
    )
    return out


# ------------------------------------------------------------------------------------------------ move() API
class _NeedsSmoothing(Exception):
    """Raised from the patched smooth_pave: the reference would smooth the mesh here (graph-based, not restated)."""


def move_inputs(seed: int, T: int):
    """Python-float (radius fraction, angle) pairs and rule selectors for BoudaryEnv.move (rl/boundary_env.py:265):
    points in the half plane where valid elements lie, selectors spread over the three branches of TYPE_THRESHOLD."""
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.uniform(0.05, 0.45, T), rng.uniform(0.2, 1.5, T)], axis=1)
    wild = rng.random(T) < 0.15
    pts[wild] = np.stack([rng.uniform(0.0, 1.2, int(wild.sum())), rng.uniform(-3.2, 3.2, int(wild.sum()))], axis=1)
    types = rng.uniform(0.0, 1.0, T)
    edge = rng.random(T) < 0.08            # the threshold values themselves
    types[edge] = rng.choice([0.3, 0.7, 1 - 0.3, 0.0, 1.0], int(edge.sum()))
    return pts.astype(np.float64), types.astype(np.float64)


def campaign_move_inputs(seed: int, T: int):
    """The parity campaign's move() inputs (tools/parity_campaign.py, oracle/check_move_vs_reference.py): radius fraction
    U(0.05, 0.45), angle U(0.2, 1.5), type U(0, 1), drawn move by move."""
    rng = np.random.default_rng(seed)
    pts, types = np.zeros((T, 2)), np.zeros(T)
    for t in range(T):
        pts[t] = (float(rng.uniform(0.05, 0.45)), float(rng.uniform(0.2, 1.5)))
        types[t] = float(rng.uniform(0, 1))
    return pts, types


def record_move_trace(points, pts: np.ndarray, types: np.ndarray, static_reset: bool = True,
                      reset_on_done: bool = True) -> dict:
    """Drive the reference's move() with Python floats and record what parity needs.  A call the reference cannot answer
    is recorded by its code: 2 = it raises UnboundLocalError (ring <= 5 on entry), 4 = it raises inside smooth_pave
    (math domain error / division by zero); `smoothed` marks the calls that went through smooth_pave (B:405-426)."""
    import contextlib
    import io
    env = make_env(points)
    n0 = len(points)
    T = len(pts)
    smoothed_flag = [0]
    orig_smooth = env.smooth_pave

    def _counting(*a, **k):
        smoothed_flag[0] = 1
        with contextlib.redirect_stdout(io.StringIO()):
            return orig_smooth(*a, **k)
    env.smooth_pave = _counting

    def ids_of(vlist):
        table = {id(v): k for k, v in enumerate(env.boundary.vertices)}
        return [table[id(v)] for v in vlist]

    reset_obs = env.reset(static=static_reset)
    out = dict(
        obs=np.zeros((T, 18), np.float32), code=np.zeros(T, np.uint8), done=np.zeros(T, np.uint8),
        complete=np.zeros(T, np.uint8), ring_len=np.zeros(T, np.int32), ring_ids=np.full((T, n0), -1, np.int32),
        ref_id=np.full(T, -1, np.int32), n_elem=np.zeros(T, np.int32), n_not_valid=np.zeros(T, np.int32),
        new_xy=np.full((T, 2), np.nan, np.float64), valid=np.zeros(T, np.uint8), was_reset=np.zeros(T, np.uint8),
        smoothed=np.zeros(T, np.uint8), warned=np.zeros(T, np.uint8),
    )
    import warnings
    for t in range(T):
        nverts_before = len(env.boundary.vertices)
        nelem_before = len(env.generated_meshes)
        smoothed_flag[0] = 0
        # `warned`: NumPy RuntimeWarnings inside the call = a zero divisor with a NumPy-scalar operand in one of the front
        # smoother's vertex constructions (the reference goes on with nan; with Python operands it raises: code 4)
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter("always")
            try:
                obs, rew, done, info = env.move([float(pts[t, 0]), float(pts[t, 1])], float(types[t]))
                assert rew == 0
                code = 1 if obs is None else 0
                comp = info["is_complete"]
            except UnboundLocalError:
                obs, done, comp, code = None, False, False, 2
            except (ValueError, ZeroDivisionError):
                assert smoothed_flag[0]
                obs, done, comp, code = None, True, False, 4
        out["warned"][t] = min(255, len(wlist))
        out["code"][t] = code
        out["smoothed"][t] = smoothed_flag[0]
        if obs is not None:
            out["obs"][t] = obs
        out["done"][t] = done
        out["complete"][t] = comp
        ring = ids_of(env.updated_boundary.vertices)
        out["ring_len"][t] = len(ring)
        out["ring_ids"][t, :len(ring)] = ring
        if code == 0:
            out["ref_id"][t] = ids_of([env.current_point_environment.reference_point])[0]
        out["n_elem"][t] = len(env.generated_meshes)
        out["n_not_valid"][t] = len(env.not_valid_points)
        out["valid"][t] = len(env.generated_meshes) > nelem_before
        if len(env.boundary.vertices) > nverts_before:
            v = env.boundary.vertices[-1]
            out["new_xy"][t] = (v.x, v.y)
        if (done and reset_on_done) or code >= 2:   # reset_on_done False: the next call on the finished ring raises (code 2)
            env.reset(static=static_reset)
            out["was_reset"][t] = 1
    out.update(domain_xy=np.array(points, np.float64), points=pts, types=types,
               reset_obs=reset_obs.astype(np.float32), static_reset=np.uint8(static_reset),
               consts=np.array([float(env.original_area), float(env.average_edge_length),
                                float(env.estimated_area_range[0]), float(env.estimated_area_range[1])], np.float64))
    return out


# ------------------------------------------------------------------------------------------------ smoothing
def record_smooth_trace(points, actions: np.ndarray, smooth_every: int, iteration: int = 400, max_v: int = 96) -> dict:
    """Drive step() and, after every `smooth_every`-th accepted element of an episode, call the reference's
    smooth_pave(boundary.vertices, updated_boundary.vertices, iteration=..., interior=True) (general/mesh.py:790-795 =
    smooth_fixed_vertices + find_reference_candidates(0); the call general/EBRD.py:393 makes on an unfinished mesh).
    Recorded per call: the step index it follows, the vertex table before and after, the sweep count (the reference only
    prints it), the element log and ring at that moment, the rebuilt candidate list.  The step records that follow show
    that stepping continues from the smoothed state exactly as the reference's does."""
    import contextlib
    import io
    import re
    env = make_env(points)
    n0 = len(points)
    T = len(actions)
    reset_obs = env.reset()

    def ids_of(vlist):
        table = {id(v): k for k, v in enumerate(env.boundary.vertices)}
        return [table[id(v)] for v in vlist]

    out = dict(
        obs=np.zeros((T, 18), np.float32), obs_none=np.zeros(T, np.uint8), reward=np.zeros(T, np.float64),
        done=np.zeros(T, np.uint8), complete=np.zeros(T, np.uint8), ring_len=np.zeros(T, np.int32),
        n_elem=np.zeros(T, np.int32), ref_id=np.full(T, -1, np.int32), valid=np.zeros(T, np.uint8),
    )
    calls = []
    since = 0
    for t in range(T):
        nelem_before = len(env.generated_meshes)
        obs, rew, done, info = env.step(actions[t])
        out["obs_none"][t] = obs is None
        if obs is not None:
            out["obs"][t] = obs
        out["reward"][t] = rew
        out["done"][t] = done
        out["complete"][t] = info["is_complete"]
        out["ring_len"][t] = len(env.updated_boundary.vertices)
        out["n_elem"][t] = len(env.generated_meshes)
        out["ref_id"][t] = ids_of([env.current_point_environment.reference_point])[0]
        valid = len(env.generated_meshes) > nelem_before
        out["valid"][t] = valid
        since += int(valid)
        if done:
            env.reset()
            since = 0
        elif valid and since >= smooth_every and len(env.boundary.vertices) <= max_v:
            since = 0
            before = np.array([(v.x, v.y) for v in env.boundary.vertices], np.float64)
            quads = np.array([ids_of(m.vertices) for m in env.generated_meshes], np.int32)
            ring = np.array(ids_of(env.updated_boundary.vertices), np.int32)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                env.smooth_pave(env.boundary.vertices, env.updated_boundary.vertices, iteration=iteration, interior=True)
            m = re.search(r"Iteration numbers: (\d+), the diff of smoothing is ([-+0-9.e]+)!", buf.getvalue())
            after = np.array([(v.x, v.y) for v in env.boundary.vertices], np.float64)
            cv = env.candidate_vertices
            calls.append(dict(t=t, before=before, after=after, quads=quads, ring=ring, sweeps=int(m.group(1)),
                              diff=float(m.group(2)), cand_ids=np.array(ids_of([c[0] for c in cv]), np.int32),
                              cand_keys=np.array([float(c[1]) for c in cv], np.float64)))
    out.update(domain_xy=np.array(points, np.float64), actions=actions, reset_obs=reset_obs.astype(np.float32),
               consts=np.array([float(env.original_area), float(env.average_edge_length),
                                float(env.estimated_area_range[0]), float(env.estimated_area_range[1])], np.float64),
               n_calls=np.int32(len(calls)), iteration=np.int32(iteration))
    # ragged per-call arrays, padded to the largest
    if calls:
        V = max(len(c["before"]) for c in calls)
        E = max(len(c["quads"]) for c in calls)
        R = max(len(c["ring"]) for c in calls)
        K = max(len(c["cand_ids"]) for c in calls)
        nC = len(calls)
        out.update(
            call_t=np.array([c["t"] for c in calls], np.int32), call_sweeps=np.array([c["sweeps"] for c in calls], np.int32),
            call_diff=np.array([c["diff"] for c in calls], np.float64),
            call_nv=np.array([len(c["before"]) for c in calls], np.int32),
            call_ne=np.array([len(c["quads"]) for c in calls], np.int32),
            call_nr=np.array([len(c["ring"]) for c in calls], np.int32),
            call_nc=np.array([len(c["cand_ids"]) for c in calls], np.int32),
            call_before=np.full((nC, V, 2), np.nan), call_after=np.full((nC, V, 2), np.nan),
            call_quads=np.full((nC, E, 4), -1, np.int32), call_ring=np.full((nC, R), -1, np.int32),
            call_cand_ids=np.full((nC, K), -1, np.int32), call_cand_keys=np.full((nC, K), np.nan),
        )
        for k, c in enumerate(calls):
            out["call_before"][k, :len(c["before"])] = c["before"]
            out["call_after"][k, :len(c["after"])] = c["after"]
            out["call_quads"][k, :len(c["quads"])] = c["quads"]
            out["call_ring"][k, :len(c["ring"])] = c["ring"]
            out["call_cand_ids"][k, :len(c["cand_ids"])] = c["cand_ids"]
            out["call_cand_keys"][k, :len(c["cand_keys"])] = c["cand_keys"]
    return out


def record_final_smooth_trace(points, actions: np.ndarray, iteration: int = 400, max_calls: int = 64) -> dict:
    """Drive step() and, whenever an episode ends COMPLETE (front of 4 or 5 vertices), call the reference's
    smooth(boundary.vertices, iteration=...) (general/mesh.py:1290-1392: the call general/EBRD.py:391 makes on a finished
    mesh) before resetting.  Recorded per call: vertex table before / after, element log, front, sweep count, and how
    many vertex visits took each of smooth()'s three branches (1 / 2 / >= 3 related meshes)."""
    import contextlib
    import io
    import re
    env = make_env(points)
    T = len(actions)
    env.reset()

    def ids_of(vlist):
        table = {id(v): k for k, v in enumerate(env.boundary.vertices)}
        return [table[id(v)] for v in vlist]

    calls = []
    done_flags = np.zeros(T, np.uint8)
    complete_flags = np.zeros(T, np.uint8)
    for t in range(T):
        obs, rew, done, info = env.step(actions[t])
        done_flags[t] = done
        complete_flags[t] = info["is_complete"]
        if done:
            if info["is_complete"] and len(calls) < max_calls:
                before = np.array([(v.x, v.y) for v in env.boundary.vertices], np.float64)
                quads = np.array([ids_of(m.vertices) for m in env.generated_meshes], np.int32)
                ring = np.array(ids_of(env.updated_boundary.vertices), np.int32)
                branch = [0, 0, 0]
                orig_frm = env.find_related_meshes

                def counting(v, _f=orig_frm, _b=branch):
                    r = _f(v)
                    _b[min(len(r), 3) - 1 if len(r) else 2] += 1
                    return r
                env.find_related_meshes = counting
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    env.smooth(env.boundary.vertices, iteration=iteration)
                env.find_related_meshes = orig_frm
                m = re.search(r"Iteration numbers: (\d+), the diff of smoothing is ([-+0-9.e]+)!", buf.getvalue())
                after = np.array([(v.x, v.y) for v in env.boundary.vertices], np.float64)
                calls.append(dict(t=t, before=before, after=after, quads=quads, ring=ring, sweeps=int(m.group(1)),
                                  diff=float(m.group(2)), branch=branch))
            env.reset()
    out = dict(domain_xy=np.array(points, np.float64), actions=actions, done=done_flags, complete=complete_flags,
               consts=np.array([float(env.original_area), float(env.average_edge_length),
                                float(env.estimated_area_range[0]), float(env.estimated_area_range[1])], np.float64),
               n_calls=np.int32(len(calls)), iteration=np.int32(iteration))
    if calls:
        V = max(len(c["before"]) for c in calls)
        E = max(len(c["quads"]) for c in calls)
        R = max(len(c["ring"]) for c in calls)
        nC = len(calls)
        out.update(
            call_t=np.array([c["t"] for c in calls], np.int32), call_sweeps=np.array([c["sweeps"] for c in calls], np.int32),
            call_diff=np.array([c["diff"] for c in calls], np.float64),
            call_branch=np.array([c["branch"] for c in calls], np.int64),
            call_nv=np.array([len(c["before"]) for c in calls], np.int32),
            call_ne=np.array([len(c["quads"]) for c in calls], np.int32),
            call_nr=np.array([len(c["ring"]) for c in calls], np.int32),
            call_before=np.full((nC, V, 2), np.nan), call_after=np.full((nC, V, 2), np.nan),
            call_quads=np.full((nC, E, 4), -1, np.int32), call_ring=np.full((nC, R), -1, np.int32),
        )
        for k, c in enumerate(calls):
            out["call_before"][k, :len(c["before"])] = c["before"]
            out["call_after"][k, :len(c["after"])] = c["after"]
            out["call_quads"][k, :len(c["quads"])] = c["quads"]
            out["call_ring"][k, :len(c["ring"])] = c["ring"]
    return out


def record_front_smooth_trace(points, actions: np.ndarray, smooth_every: int, iteration: int = 400) -> dict:
    """As record_smooth_trace, but the call is smooth_pave(..., interior=False) -- smooth_current_boundary_3 on the front
    (general/mesh.py:939-1028), then the interior relaxation and the candidate rebuild -- followed by the
    find_next_state() that move() runs right after it (rl/boundary_env.py:405-420), whose observation is recorded.  A
    call in which the reference raises (math domain error / division by zero inside the vertex constructions) is
    recorded with raised = 1 and the partly smoothed vertex table; the episode is reset after it."""
    import contextlib
    import io
    import re
    env = make_env(points)
    T = len(actions)
    reset_obs = env.reset()

    def ids_of(vlist):
        table = {id(v): k for k, v in enumerate(env.boundary.vertices)}
        return [table[id(v)] for v in vlist]

    out = dict(
        obs=np.zeros((T, 18), np.float32), obs_none=np.zeros(T, np.uint8), reward=np.zeros(T, np.float64),
        done=np.zeros(T, np.uint8), complete=np.zeros(T, np.uint8), ring_len=np.zeros(T, np.int32),
        n_elem=np.zeros(T, np.int32), valid=np.zeros(T, np.uint8),
    )
    calls = []
    since = 0
    for t in range(T):
        nelem_before = len(env.generated_meshes)
        obs, rew, done, info = env.step(actions[t])
        out["obs_none"][t] = obs is None
        if obs is not None:
            out["obs"][t] = obs
        out["reward"][t] = rew
        out["done"][t] = done
        out["complete"][t] = info["is_complete"]
        out["ring_len"][t] = len(env.updated_boundary.vertices)
        out["n_elem"][t] = len(env.generated_meshes)
        valid = len(env.generated_meshes) > nelem_before
        out["valid"][t] = valid
        since += int(valid)
        if done:
            env.reset()
            since = 0
        elif valid and since >= smooth_every:
            since = 0
            before = np.array([(v.x, v.y) for v in env.boundary.vertices], np.float64)
            quads = np.array([ids_of(m.vertices) for m in env.generated_meshes], np.int32)
            ring = np.array(ids_of(env.updated_boundary.vertices), np.int32)
            buf = io.StringIO()
            raised, sweeps, s_obs, s_none = 0, -1, np.zeros(18, np.float32), 0
            try:
                with contextlib.redirect_stdout(buf):
                    env.smooth_pave(env.boundary.vertices, env.updated_boundary.vertices, iteration=iteration, interior=False)
                m = re.search(r"Iteration numbers: (\d+), the diff of smoothing is ([-+0-9.e]+)!", buf.getvalue())
                sweeps = int(m.group(1))
                o2 = env.find_next_state(env.not_valid_points)
                s_none = o2 is None
                if o2 is not None:
                    s_obs = np.asarray(o2, np.float32)
            except (ValueError, ZeroDivisionError):
                raised = 1
            after = np.array([(v.x, v.y) for v in env.boundary.vertices], np.float64)
            cv = env.candidate_vertices
            calls.append(dict(t=t, before=before, after=after, quads=quads, ring=ring, sweeps=sweeps, raised=raised,
                              obs=s_obs, obs_none=s_none,
                              ref=-1 if (raised or s_none) else ids_of([env.current_point_environment.reference_point])[0],
                              cand_ids=np.array(ids_of([c[0] for c in cv]), np.int32),
                              cand_keys=np.array([float(c[1]) for c in cv], np.float64)))
            if raised:
                env.reset()
    out.update(domain_xy=np.array(points, np.float64), actions=actions, reset_obs=reset_obs.astype(np.float32),
               consts=np.array([float(env.original_area), float(env.average_edge_length),
                                float(env.estimated_area_range[0]), float(env.estimated_area_range[1])], np.float64),
               n_calls=np.int32(len(calls)), iteration=np.int32(iteration))
    if calls:
        V = max(len(c["before"]) for c in calls)
        E = max(len(c["quads"]) for c in calls)
        R = max(len(c["ring"]) for c in calls)
        K = max(len(c["cand_ids"]) for c in calls)
        nC = len(calls)
        out.update(
            call_t=np.array([c["t"] for c in calls], np.int32), call_sweeps=np.array([c["sweeps"] for c in calls], np.int32),
            call_raised=np.array([c["raised"] for c in calls], np.uint8),
            call_obs=np.stack([c["obs"] for c in calls]), call_obs_none=np.array([c["obs_none"] for c in calls], np.uint8),
            call_ref=np.array([c["ref"] for c in calls], np.int32),
            call_nv=np.array([len(c["before"]) for c in calls], np.int32),
            call_ne=np.array([len(c["quads"]) for c in calls], np.int32),
            call_nr=np.array([len(c["ring"]) for c in calls], np.int32),
            call_nc=np.array([len(c["cand_ids"]) for c in calls], np.int32),
            call_before=np.full((nC, V, 2), np.nan), call_after=np.full((nC, V, 2), np.nan),
            call_quads=np.full((nC, E, 4), -1, np.int32), call_ring=np.full((nC, R), -1, np.int32),
            call_cand_ids=np.full((nC, K), -1, np.int32), call_cand_keys=np.full((nC, K), np.nan),
        )
        for k, c in enumerate(calls):
            out["call_before"][k, :len(c["before"])] = c["before"]
            out["call_after"][k, :len(c["after"])] = c["after"]
            out["call_quads"][k, :len(c["quads"])] = c["quads"]
            out["call_ring"][k, :len(c["ring"])] = c["ring"]
            out["call_cand_ids"][k, :len(c["cand_ids"])] = c["cand_ids"]
            out["call_cand_keys"][k, :len(c["cand_keys"])] = c["cand_keys"]
    return out
