"""Replay of a smooth_*.npz fixture (step() with smooth_pave(interior=True) calls in between) against any implementation.

The fixture's records were produced by the reference itself (oracle/ref_harness.record_smooth_trace); `impl` adapts the
oracle (tests/test_oracle_smooth.py) or the HIP path (tests/test_gpu_smooth.py):

    impl.reset() -> obs[18]
    impl.step(a[3]) -> obs[18], reward, done, complete           (finished episodes are reset by the replay)
    impl.smooth(iteration) -> sweeps
    impl.vertices() -> xy[n_vert, 2]      impl.elements() -> quads[n_elem, 4]
    impl.ring_ids() -> ids[n]             impl.candidates() -> (ids, keys) in the reference's list order
"""
import numpy as np


def replay(tr, impl, obs_tol=0.0, exact_vertices=True, key_tol=0.0):
    T = len(tr["actions"])
    calls = {int(t): k for k, t in enumerate(tr["call_t"])} if int(tr["n_calls"]) else {}
    iteration = int(tr["iteration"])
    o = impl.reset()
    assert np.abs(o.astype(np.float64) - tr["reset_obs"]).max() <= obs_tol
    moved_total = 0
    for t in range(T):
        o, r, d, c = impl.step(tr["actions"][t])
        assert bool(d) == bool(tr["done"][t]) and bool(c) == bool(tr["complete"][t]), t
        if not tr["obs_none"][t]:
            assert np.abs(np.asarray(o, np.float64) - tr["obs"][t]).max() <= max(obs_tol, 0.0), t
        assert abs(r - tr["reward"][t]) <= max(obs_tol, 1e-12), t
        if d:
            impl.reset()
        if t in calls:
            k = calls[t]
            nv, ne, nr, nc = (int(tr[x][k]) for x in ("call_nv", "call_ne", "call_nr", "call_nc"))
            # the state the reference smoothed is the state we are in
            assert np.array_equal(impl.elements(), tr["call_quads"][k, :ne]), t
            assert np.array_equal(impl.ring_ids(), tr["call_ring"][k, :nr]), t
            assert np.array_equal(impl.vertices(), tr["call_before"][k, :nv]), t
            sweeps = impl.smooth(iteration)
            assert sweeps == int(tr["call_sweeps"][k]), (t, sweeps, int(tr["call_sweeps"][k]))
            got, want = impl.vertices(), tr["call_after"][k, :nv]
            if exact_vertices:
                assert np.array_equal(got, want), (t, np.abs(got - want).max())
            else:
                assert np.abs(got - want).max() <= 1e-12, t
            moved_total += int(np.any(want != tr["call_before"][k, :nv], axis=1).sum())
            ids, keys = impl.candidates()
            assert np.array_equal(ids, tr["call_cand_ids"][k, :nc]), t
            assert np.abs(keys - tr["call_cand_keys"][k, :nc]).max(initial=0.0) <= key_tol, t
    return moved_total


def replay_final(tr, impl, vertex_tol=0.0):
    """smoothfinal_*.npz: step() until an episode ends complete, then smooth() (general/mesh.py:1290-1392) before the reset.
    impl as above plus impl.smooth_final(iteration) -> sweeps.  Returns the largest vertex deviation seen."""
    calls = {int(t): k for k, t in enumerate(tr["call_t"])}
    iteration = int(tr["iteration"])
    impl.reset()
    worst = 0.0
    for t in range(len(tr["actions"])):
        o, r, d, c = impl.step(tr["actions"][t])
        assert bool(d) == bool(tr["done"][t]) and bool(c) == bool(tr["complete"][t]), t
        if t in calls:
            k = calls[t]
            nv, ne, nr = (int(tr[x][k]) for x in ("call_nv", "call_ne", "call_nr"))
            assert np.array_equal(impl.elements(), tr["call_quads"][k, :ne]), t
            assert np.array_equal(impl.ring_ids(), tr["call_ring"][k, :nr]), t
            assert np.array_equal(impl.vertices(), tr["call_before"][k, :nv]), t
            sweeps = impl.smooth_final(iteration)
            assert sweeps == int(tr["call_sweeps"][k]), (t, sweeps, int(tr["call_sweeps"][k]))
            dev = float(np.abs(impl.vertices() - tr["call_after"][k, :nv]).max())
            assert dev <= vertex_tol, (t, dev)
            worst = max(worst, dev)
        if d:
            impl.reset()
    return worst


def replay_front(tr, impl, obs_tol=0.0, vertex_tol=0.0, key_tol=0.0):
    """smoothfront_*.npz: step() with smooth_pave(interior=False) + find_next_state calls in between.
    impl.smooth_full(iteration) -> (code, sweeps, obs): code 0 / 1 observation array / None, -3 the reference raises."""
    calls = {int(t): k for k, t in enumerate(tr["call_t"])}
    iteration = int(tr["iteration"])
    o = impl.reset()
    assert np.abs(o.astype(np.float64) - tr["reset_obs"]).max() <= obs_tol
    worst = 0.0
    front_moves = 0
    for t in range(len(tr["actions"])):
        o, r, d, c = impl.step(tr["actions"][t])
        assert bool(d) == bool(tr["done"][t]) and bool(c) == bool(tr["complete"][t]), t
        if not tr["obs_none"][t]:
            assert np.abs(np.asarray(o, np.float64) - tr["obs"][t]).max() <= obs_tol, t
        assert abs(r - tr["reward"][t]) <= max(obs_tol, 1e-12), t
        if d:
            impl.reset()
        if t in calls:
            k = calls[t]
            nv, ne, nr, nc = (int(tr[x][k]) for x in ("call_nv", "call_ne", "call_nr", "call_nc"))
            assert np.array_equal(impl.elements(), tr["call_quads"][k, :ne]), t
            assert np.array_equal(impl.ring_ids(), tr["call_ring"][k, :nr]), t
            assert np.abs(impl.vertices() - tr["call_before"][k, :nv]).max() <= vertex_tol, t
            code, sweeps, obs = impl.smooth_full(iteration)
            want = tr["call_after"][k, :nv]
            dev = float(np.abs(impl.vertices() - want).max())
            assert dev <= vertex_tol, (t, dev)
            worst = max(worst, dev)
            ring = tr["call_ring"][k, :nr]
            front_moves += int(np.any(want[ring] != tr["call_before"][k, :nv][ring], axis=1).sum())
            if tr["call_raised"][k]:
                assert code == -3, (t, code)
                impl.reset()
                continue
            assert code == int(tr["call_obs_none"][k]), (t, code)
            assert sweeps == int(tr["call_sweeps"][k]), (t, sweeps, int(tr["call_sweeps"][k]))
            if not tr["call_obs_none"][k]:
                assert np.abs(np.asarray(obs, np.float64) - tr["call_obs"][k]).max() <= obs_tol, t
                assert impl.ref_id() == int(tr["call_ref"][k]), t
            ids, keys = impl.candidates()
            assert np.array_equal(ids, tr["call_cand_ids"][k, :nc]), t
            assert np.abs(keys - tr["call_cand_keys"][k, :nc]).max(initial=0.0) <= key_tol, t
    return worst, front_moves
