"""Stand-ins for gymnasium / gym / stable_baselines3 (none of them is in the image), registered in sys.modules the way the
reference's own tests do it (v2/tests/mesh_rl/test_boundary_env_equiv.py:17-151).  They carry what Stable-Baselines3
checks before it accepts an environment: the ``Env`` base class, ``spaces.Box``, and a ``VecEnv`` abstract base with the
abstract methods and the constructor bookkeeping of SB3 2.x (``reset_infos``, ``_seeds``, ``get_attr("render_mode")``),
plus ``wrap_env`` = the test SB3's ``BaseAlgorithm._wrap_env`` applies."""
import abc
import sys
import types

import numpy as np


def install(with_gymnasium=True, with_gym=False, with_sb3=True):
    """Registers the stubs; returns the dict of module names -> previous sys.modules entries for uninstall()."""
    saved = {}

    def put(name, mod):
        saved[name] = sys.modules.get(name)
        sys.modules[name] = mod

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            if shape is not None:
                self.low, self.high = np.full(tuple(shape), low, dtype=dtype), np.full(tuple(shape), high, dtype=dtype)
            else:
                self.low, self.high = np.asarray(low, dtype=dtype), np.asarray(high, dtype=dtype)
            self.shape, self.dtype = self.low.shape, np.dtype(dtype)
            self._rng = np.random.default_rng(0)

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

        def sample(self):
            return self._rng.uniform(self.low, self.high).astype(self.dtype)

    for name, on in (("gymnasium", with_gymnasium), ("gym", with_gym)):
        if not on:
            continue
        mod, spaces = types.ModuleType(name), types.ModuleType(name + ".spaces")

        class Env:
            metadata = {}
            render_mode = None
            _np_random = None

            def reset(self, *, seed=None, options=None):
                if seed is not None:
                    self._np_random = np.random.default_rng(seed)

            @property
            def unwrapped(self):
                return self

        Env.__module__ = name
        spaces.Box = Box
        mod.Env, mod.spaces = Env, spaces
        put(name, mod)
        put(name + ".spaces", spaces)

    if with_sb3:
        sb3 = types.ModuleType("stable_baselines3")
        common = types.ModuleType("stable_baselines3.common")
        vec = types.ModuleType("stable_baselines3.common.vec_env")

        class VecEnv(abc.ABC):
            def __init__(self, num_envs, observation_space, action_space):
                self.num_envs = num_envs
                self.observation_space = observation_space
                self.action_space = action_space
                self.reset_infos = [{} for _ in range(num_envs)]
                self._seeds = [None for _ in range(num_envs)]
                self._options = [{} for _ in range(num_envs)]
                render_modes = self.get_attr("render_mode")
                assert all(m == render_modes[0] for m in render_modes)
                self.render_mode = render_modes[0]

            def _reset_seeds(self):
                self._seeds = [None for _ in range(self.num_envs)]

            def _reset_options(self):
                self._options = [{} for _ in range(self.num_envs)]

            @abc.abstractmethod
            def reset(self): ...

            @abc.abstractmethod
            def step_async(self, actions): ...

            @abc.abstractmethod
            def step_wait(self): ...

            @abc.abstractmethod
            def close(self): ...

            @abc.abstractmethod
            def get_attr(self, attr_name, indices=None): ...

            @abc.abstractmethod
            def set_attr(self, attr_name, value, indices=None): ...

            @abc.abstractmethod
            def env_method(self, method_name, *method_args, indices=None, **method_kwargs): ...

            @abc.abstractmethod
            def env_is_wrapped(self, wrapper_class, indices=None): ...

            def step(self, actions):
                self.step_async(actions)
                return self.step_wait()

        vec.VecEnv = VecEnv
        common.vec_env = vec
        sb3.common = common
        put("stable_baselines3", sb3)
        put("stable_baselines3.common", common)
        put("stable_baselines3.common.vec_env", vec)
    return saved


def uninstall(saved):
    for name, old in saved.items():
        if old is None:
            sys.modules.pop(name, None)
        else:
            sys.modules[name] = old


def wrap_env(env):
    """What BaseAlgorithm._wrap_env decides: a VecEnv is taken as it is; anything else must be a gym(nasium) Env and is
    put into a DummyVecEnv.  Returns "vecenv" / "env"; raises TypeError for an object SB3 would refuse."""
    vec = sys.modules["stable_baselines3.common.vec_env"].VecEnv
    if isinstance(env, vec):
        return "vecenv"
    for name in ("gymnasium", "gym"):
        if name in sys.modules and isinstance(env, sys.modules[name].Env):
            return "env"
    raise TypeError(f"{type(env).__name__} is neither a VecEnv nor a gym Env")
