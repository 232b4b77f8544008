"""GPU: the drop-in boundary as Stable-Baselines3 sees it.  gymnasium / stable_baselines3 are not in the image, so stub modules
(tests/rl_stubs.py, the reference's own technique: v2/tests/mesh_rl/test_boundary_env_equiv.py:17-151) provide the base
classes; the environments behind them are the real HIP ones.  Covers: SB3's isinstance gate (`_wrap_env`), the Gymnasium
call shapes, the VecEnv contract of rl/baselines/dummy_vec_env.py:12-125, and the evaluation callback's exact
`env.envs[0].save_meshes(..., meshes=env.envs[0].generated_meshes, indexing=True, style='k-', dpi=30)` call
(rl/baselines/CustomizeCallback.py:131-133) -- its PNG compared with the one the reference wrote for the same episode."""
import importlib
import io
import os
import sys

import numpy as np
import pytest

import rl_stubs
from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu
PKG = "reinforcementlearning4meshgeneration_amd"


def _reload():
    for m in ("compat", "episode_tools", "vec_env", "boundary_env"):
        importlib.reload(importlib.import_module(f"{PKG}.{m}"))
    return sys.modules[f"{PKG}.vec_env"], sys.modules[f"{PKG}.boundary_env"]


@pytest.fixture
def rl_stack():
    saved = rl_stubs.install(with_gymnasium=True, with_gym=False, with_sb3=True)
    try:
        yield _reload()
    finally:
        rl_stubs.uninstall(saved)
        _reload()


def _pixels(png_bytes):
    import matplotlib.image as mpimg
    return mpimg.imread(io.BytesIO(bytes(png_bytes)), format="png")


def _replay_to_finished_episode(step_fn, reset_fn, fx):
    tr = dict(np.load(os.path.join(GOLDEN_DIR, str(fx["trace"]) + ".npz")))
    reset_fn()
    for t in range(int(fx["step"]) + 1):
        done = step_fn(tr["actions"][t])
        if done and t < int(fx["step"]):
            reset_fn()
    assert done
    return tr


def test_boudaryenv_is_a_gymnasium_env_and_draws_the_reference_png(rl_stack, tmp_path):
    import matplotlib
    matplotlib.use("Agg")
    vec_env, boundary_env = rl_stack
    from reinforcementlearning4meshgeneration_amd.domains import boundary
    fx = dict(np.load(os.path.join(GOLDEN_DIR, "plot_boundary0_biased_s1.npz")))
    env = boundary_env.BoudaryEnv(boundary(0))
    assert isinstance(env, sys.modules["gymnasium"].Env) and rl_stubs.wrap_env(env) == "env"
    assert env.api == "gymnasium" and env.render_mode is None and env.unwrapped is env
    obs, info = env.reset(seed=3)                      # Gymnasium call shapes, v2/src/mesh_rl/envs/boundary_env.py:136-184
    assert obs.shape == (18,) and obs.dtype == np.float32 and info == {}
    out = env.step(env.action_space.sample())
    assert len(out) == 5 and isinstance(out[2], bool) and isinstance(out[3], bool) and set(out[4]) == {"is_complete"}

    def step(a):
        _, _, term, trunc, _ = env.step(a)
        return term or trunc
    _replay_to_finished_episode(step, env.reset, fx)
    meshes = env.generated_meshes
    assert len(meshes) == len(fx["quads"])
    np.testing.assert_array_equal(np.stack(meshes), fx["vertex_xy"][fx["quads"]])
    # rl/baselines/testbed.py:194-197: env.get_quality(env.generated_meshes[i], 4); every index that depends on the quad alone
    for c, index in enumerate(fx["quality_index"]):
        got = np.array([env.get_quality(m, int(index)) for m in meshes[:12]])
        np.testing.assert_allclose(got, fx["quality"][:12, c], rtol=0, atol=1e-12, err_msg=f"get_quality(element, {index})")
    with pytest.raises(ValueError):
        env.get_quality(meshes[0], 2)
    same_version = matplotlib.__version__ == str(fx["matplotlib_version"])
    calls = {"callback": dict(indexing=True, style="k-", dpi=30),
             "testbed": dict(quality=False, type=4, indexing=False, style="k-", dpi=40),
             "labelled": dict(quality=True, indexing=True, type=4, dpi=40)}       # the labels' numbers come from the device
    for key, kw in calls.items():
        path = tmp_path / f"{key}.png"
        env.save_meshes(str(path), meshes=env.generated_meshes, **kw)
        assert path.exists() and path.stat().st_size > 1000
        if same_version:
            got, want = _pixels(open(path, "rb").read()), _pixels(fx["png_" + key])
            assert got.shape == want.shape and np.array_equal(got, want), key
    # testbed.py:216-219: extract_samples_2 -> save_samples(..., _type=2)
    samples, types, outputs = env.extract_samples_2(env.generated_meshes, 2, 3, radius=4)
    env.save_samples(str(tmp_path / "s.json"), {"samples": samples, "output_types": types, "outputs": outputs}, _type=2)
    import json
    back = json.load(open(tmp_path / "s.json"))
    assert len(back["samples"]) == len(samples) > 0 and len(back["samples"][0]) == 14
    env.close()


def test_sb3_vecenv_contract_and_the_callbacks_save_meshes_call(rl_stack, tmp_path):
    import matplotlib
    matplotlib.use("Agg")
    vec_env, _ = rl_stack
    from reinforcementlearning4meshgeneration_amd.domains import boundary
    VecEnv = sys.modules["stable_baselines3.common.vec_env"].VecEnv
    fx = dict(np.load(os.path.join(GOLDEN_DIR, "plot_boundary0_biased_s1.npz")))
    # (1) the evaluation env of the callback: one env, no auto-reset (rl/baselines/dummy_vec_env.py:49)
    ev = vec_env.SB3MeshVecEnv([boundary(0)], n_envs=1, auto_reset=False, log_capacity=512)
    assert isinstance(ev, VecEnv) and rl_stubs.wrap_env(ev) == "vecenv" and ev.num_envs == 1 and ev.render_mode is None
    assert len(ev.envs) == 1 and ev.reset_infos == [{}]
    obs = ev.reset()
    assert isinstance(obs, np.ndarray) and obs.shape == (1, 18) and obs.dtype == np.float32

    def step(a):
        o, r, d, infos = ev.step(a[None])
        assert o.shape == (1, 18) and r.dtype == np.float32 and d.dtype == bool and isinstance(infos[0], dict)
        return bool(d[0])
    _replay_to_finished_episode(step, ev.reset, fx)
    path = tmp_path / "eval.png"
    ev.envs[0].save_meshes(str(path), meshes=ev.envs[0].generated_meshes, indexing=True, style='k-', dpi=30)   # CustomizeCallback.py:131-133
    assert path.exists()
    if matplotlib.__version__ == str(fx["matplotlib_version"]):
        assert np.array_equal(_pixels(open(path, "rb").read()), _pixels(fx["png_callback"]))
    assert len(ev.get_attr("generated_meshes")[0]) == len(fx["quads"])
    q4 = ev.env_method("get_quality", ev.envs[0].generated_meshes[0], 4)
    np.testing.assert_allclose(q4, [fx["quality"][0, 3]], rtol=0, atol=1e-12)
    ev.close()
    # (2) the training env: many envs, auto-reset, infos per env
    n = 256
    tv = vec_env.SB3MeshVecEnv([boundary(0)], n_envs=n)
    assert isinstance(tv, VecEnv) and tv.num_envs == n and tv.seed(7) == [7] * n
    obs = tv.reset()
    assert obs.shape == (n, 18) and len(tv.reset_infos) == n
    rng = np.random.default_rng(0)
    dones = 0
    for _ in range(140):
        a = rng.uniform([-1, -1.5, 0], [1, 1.5, 1.5], size=(n, 3)).astype(np.float32)
        obs, rew, done, infos = tv.step(a)
        assert obs.shape == (n, 18) and rew.shape == (n,) and done.shape == (n,) and len(infos) == n
        for k in np.nonzero(done)[0]:
            assert infos[k]["terminal_observation"].shape == (18,) and "TimeLimit.truncated" in infos[k]
            dones += 1
        infos[0]["episode"] = {"r": 0.0}          # Monitor-style writes must not leak into other envs' dicts
        assert "episode" not in infos[1]
    assert dones > 0
    assert tv.get_attr("render_mode", indices=[0, 5]) == [None, None] and tv.env_is_wrapped(object) == [False] * n
    tv.close()
    # under auto-reset the finished mesh is the ARCHIVED episode: envs[k].last_generated_meshes / save_meshes(which="last")
    av = vec_env.SB3MeshVecEnv([boundary(0)], n_envs=64, log_capacity=256)
    av.reset()
    finished = None
    for _ in range(1500):
        a = np.stack([rng.uniform(-0.49, 0.49, 64), rng.uniform(0.2, 1.0, 64), rng.uniform(0.3, 1.2, 64)], 1).astype(np.float32)
        _, _, done, infos = av.step(a)
        hit = [k for k in np.nonzero(done)[0] if len(av.envs[int(k)].last_generated_meshes) > 5]   # complete or truncated
        if hit:
            finished = int(hit[0])
            break
    assert finished is not None
    view = av.envs[finished]
    last = view.last_generated_meshes
    assert len(last) > 5 and len(view.generated_meshes) == 0          # the running episode has just been reset
    path = tmp_path / "last.png"
    view.save_meshes(str(path), meshes=last, indexing=True, style='k-', dpi=30, which="last")
    assert path.exists() and path.stat().st_size > 1000
    av.close()
    # (3) the alternative constructors build the same kind of object (BASELINE configs[4]: one generated ring per env)
    rv = vec_env.SB3MeshVecEnv.from_random(64, 1000)
    assert isinstance(rv, VecEnv) and rv.num_envs == 64 and rv.lazy_infos is False and rv.reset().shape == (64, 18)
    obs, rew, done, infos = rv.step(rng.uniform([-1, -1.5, 0], [1, 1.5, 1.5], size=(64, 3)).astype(np.float32))
    assert obs.shape == (64, 18) and len(infos) == 64 and infos[0] is not infos[1]
    rv.close()
