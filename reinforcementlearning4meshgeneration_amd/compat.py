"""Base classes of the RL stack the reference plugs into, resolved once at import time.

The reference's env is a ``gym.Env`` (rl/boundary_env.py:18; v2/src/mesh_rl/envs/boundary_env.py:34 subclasses
``gymnasium.Env``) and its vectorised caller a Stable-Baselines3 ``VecEnv`` (rl/baselines/dummy_vec_env.py:12).  SB3
tests ``isinstance(env, gymnasium.Env)`` / ``isinstance(env, VecEnv)`` before it wraps anything
(``BaseAlgorithm._wrap_env``), so the drop-in classes must carry those bases whenever the libraries are importable.
When they are not (this image ships neither), the base is ``object`` and nothing else changes.
"""
from __future__ import annotations


def _resolve_env_base():
    """(base class, flavour): gymnasium.Env -> gym.Env -> object."""
    for name in ("gymnasium", "gym"):
        try:
            mod = __import__(name)
            base = mod.Env
        except Exception:
            continue
        if isinstance(base, type):
            return base, name
    return object, None


def _resolve_vec_env_base():
    """stable_baselines3's VecEnv, or object."""
    try:
        from stable_baselines3.common.vec_env import VecEnv
    except Exception:
        try:
            from stable_baselines3.common.vec_env.base_vec_env import VecEnv
        except Exception:
            return object
    return VecEnv if isinstance(VecEnv, type) else object


ENV_BASE, ENV_FLAVOUR = _resolve_env_base()
VEC_ENV_BASE = _resolve_vec_env_base()
