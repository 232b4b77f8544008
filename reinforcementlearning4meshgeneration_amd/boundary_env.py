"""Drop-in Gym surface of the reference's ``BoudaryEnv`` (rl/boundary_env.py:18) over the HIP engine.

``BoudaryEnv(boundary)`` keeps the reference's constructor, attributes and return conventions:

* legacy API (rl/boundary_env.py):  ``reset(static=False) -> obs``;
  ``step(action) -> (obs | None, np.float64 reward, bool done, {'is_complete': bool})``
* Gymnasium API (v2/src/mesh_rl/envs/boundary_env.py:136-184, 388-457), selected with ``api="gymnasium"``:
  ``reset(*, seed=None, static=False, options=None) -> (obs, {})``;
  ``step(action) -> (obs, reward, terminated, truncated, info)`` with
  ``terminated = done and is_complete``, ``truncated = done and not is_complete``.

The class IS a ``gymnasium.Env`` (or a ``gym.Env`` where only gym is installed) whenever one of the two is importable
(compat.ENV_BASE): Stable-Baselines3 tests ``isinstance(env, gym.Env)`` before it wraps an env in its DummyVecEnv, which
is what ``SAC('MlpPolicy', env, ...)`` of rl/baselines/RL_Mesh.py:186-197 relies on.  ``api`` defaults to the flavour of
that base -- Gymnasium 5-tuples under gymnasium (what SB3 >= 2 calls), the reference's 4-tuples otherwise.

It is a one-environment ``MeshVecEnv``: every call is a kernel launch plus a device->host copy, so it exists
for API compatibility (SB3 wraps it in its own DummyVecEnv) and for evaluation scripts; training at scale
should hand ``SB3MeshVecEnv`` to SB3 directly.  There is no CPU path: without the GPU library it raises.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

from .compat import ENV_BASE, ENV_FLAVOUR
from .domains import Point, read_polygon
from .episode_tools import EpisodeTools
from .vec_env import MeshVecEnv, make_spaces, smoothing_log_capacity


class BoudaryEnv(EpisodeTools, ENV_BASE):  # the reference's spelling; rl/boundary_env.py:18 `class BoudaryEnv(MeshGeneration, gym.Env)`
    # log_capacity: elements / generated vertices kept per episode (`generated_meshes`).  Default: the largest value (up to
    # 4096) for which the smoothers' per-env graph still fits one CU's 160 KB of LDS for this domain's ring
    # (vec_env.smoothing_log_capacity: ~2700 for boundary(), ~2400 for the 272-vertex d3 ring); an episode that outgrows it
    # keeps exact counts and rewards, but `generated_meshes` / exports are truncated (a RuntimeWarning says so) and
    # smooth() / smooth_pave() refuse it.
    metadata = {"render.modes": ["human"], "render_modes": ["human"]}
    TYPE_THRESHOLD = 0.3
    render_mode = None

    def __init__(self, boundary: Sequence[Point], experiment_version=None, env_name=None, *, device: int = 0,
                 api: Optional[str] = None, log_capacity: Optional[int] = None):
        if hasattr(boundary, "vertices"):  # a reference-style Boundary2D
            boundary = [(v.x, v.y) for v in boundary.vertices]
        if api is None:
            api = "gymnasium" if ENV_FLAVOUR == "gymnasium" else "legacy"
        if api not in ("legacy", "gymnasium"):
            raise ValueError("api must be 'legacy' or 'gymnasium'")
        self.api = api
        self.points = [tuple(p) for p in boundary]
        self.experiment_version = experiment_version if experiment_version else "test"
        self.env_name = env_name if env_name is not None else 1
        self.observation_space, self.action_space = make_spaces()
        if log_capacity is None:
            log_capacity = smoothing_log_capacity(len(self.points))
        self._vec = MeshVecEnv([self.points], n_envs=1, device=device, auto_reset=False, log_capacity=log_capacity)
        self.original_area = self._vec.constants[0].original_area
        self.average_edge_length = self._vec.constants[0].average_edge_length
        self.estimated_area_range = (self._vec.constants[0].est_min_l, self._vec.constants[0].est_crit_l)
        self.neighbor_num, self.radius_num, self.radius = 6, 3, 4
        self.history_info = {-1: [], 1: [], 0: []}   # rl/boundary_env.py:61-65: reward of every accepted element, per rule
        self._valid_seen = 0
        self.current_state = None

    @classmethod
    def from_domain_file(cls, path, **kw):
        """v2/src/mesh_rl/envs/boundary_env.py:45-56"""
        return cls(read_polygon(path), **kw)

    # ------------------------------------------------------------------ Gym surface
    def reset(self, static=False, *, seed=None, options=None):
        """rl/boundary_env.py:67 `reset(static=False)` (positional, as the legacy callers pass it) and the Gymnasium
        keywords of v2/src/mesh_rl/envs/boundary_env.py:136."""
        if seed is not None and ENV_FLAVOUR == "gymnasium":
            super().reset(seed=seed)          # gymnasium.Env.reset seeds self.np_random
        if seed is not None and hasattr(self.action_space, "seed"):
            self.action_space.seed(seed)
        obs = self._vec.reset(static=bool(static)).cpu().numpy()[0].copy()
        self.current_state = obs
        return (obs, {}) if self.api == "gymnasium" else obs

    def step(self, action):
        import torch
        a = torch.as_tensor(np.asarray(action, dtype=np.float32).reshape(1, 3), device=self._vec.device)
        obs, rew, done, comp = self._vec.step(a)
        none = bool(int(self._vec.status().cpu()[0]) & 1)
        obs_np = None if none else obs.cpu().numpy()[0].copy()
        reward = np.float64(rew.cpu()[0].item())
        done_b, comp_b = bool(done.cpu()[0]), bool(comp.cpu()[0])
        self.current_state = obs_np
        # rl/boundary_env.py:229: `self.history_info[rule].append(reward)` on an accepted element, before the +10 of a
        # finished ring is added (:232); rule by the thresholds of :144-186 (a rule-0 action that reuses an existing vertex
        # stays rule 0).  The +10 is taken off again here, which can cost the last bit of that one entry.
        valid_now = self._vec.counters()["valid"]
        if valid_now > self._valid_seen:
            a0 = float(np.asarray(action, dtype=np.float32).reshape(-1)[0])
            rule = -1 if a0 <= -0.5 else (1 if a0 >= 0.5 else 0)
            self.history_info[rule].append(np.float64(reward - 10) if (done_b and comp_b) else reward)
        self._valid_seen = valid_now
        info = {"is_complete": comp_b}
        if self.api == "gymnasium":
            return obs_np, reward, done_b and comp_b, done_b and not comp_b, info
        return obs_np, reward, done_b, info

    def move(self, new_point, type, lr_1=None, lr_2=None):
        """rl/boundary_env.py:265-432: deterministic extraction driven by (radius fraction, angle) and a rule selector;
        returns (obs | None, 0, done, {'is_complete': bool}) like the reference, including its two exceptions: on a
        finished ring (<= 5 vertices) it leaves `is_complete` unbound and raises UnboundLocalError -- so does this; and
        where it raises inside the smooth_pave it runs when no reference vertex is selectable (math domain error /
        division by zero in a vertex construction) this raises ValueError.  That smooth_pave itself (front + interior
        smoothing, candidate rebuild, the last_not_valid_points check) runs on the device inside the same call."""
        import torch

        from . import _capi
        pts = torch.tensor([[float(new_point[0]), float(new_point[1])]], dtype=torch.float64, device=self._vec.device)
        typ = torch.tensor([float(type)], dtype=torch.float64, device=self._vec.device)
        obs, done, comp, code = self._vec.move(pts, typ)
        code = int(code.cpu()[0])
        if code == _capi.MOVE_RAISES:
            raise UnboundLocalError("local variable 'is_complete' referenced before assignment "
                                    "(move() on a ring of <= 5 vertices, as in the reference)")
        if code == _capi.MOVE_SMOOTH_RAISES:
            raise ValueError("math domain error / division by zero inside smooth_pave, as in the reference")
        obs_np = obs.cpu().numpy()[0].copy() if code == _capi.MOVE_OK else None
        self.current_state = obs_np
        return obs_np, 0, bool(done.cpu()[0]), {"is_complete": bool(comp.cpu()[0])}

    def smooth_pave(self, vertices=None, current_boundary_vertices=None, lr_1=None, lr_2=None, iteration=400,
                    interior=False):
        """MeshGeneration.smooth_pave, general/mesh.py:790-795.  The two vertex lists of the reference's signature are
        implied here (boundary.vertices and the current front of this env) and ignored; lr_1 / lr_2 are unused by the
        reference as well.  interior=True (the call general/EBRD.py:393 makes): the generated vertices off the front are
        relaxed and the candidate list is rebuilt.  interior=False: smooth_current_boundary_3 moves the front's generated
        vertices first; since the front moved, the point environment is recomputed right away (the find_next_state that
        move() runs after smooth_pave) and `current_state` holds its observation.  Returns the number of sweeps (the
        reference prints it); raises ValueError where the reference raises inside the front smoother."""
        from . import _capi
        sweeps, _ = self._vec.smooth_pave(iteration=iteration, interior=interior)
        n = int(sweeps.cpu()[0])
        if n == _capi.SMOOTH_RAISES:
            raise ValueError("math domain error / division by zero inside smooth_current_boundary_3, as in the reference")
        if n == _capi.SMOOTH_NONFINITE:
            # the reference accepts the nan position here and raises in the find_next_state its callers run next
            raise ValueError("cannot convert float NaN to integer (a NaN vertex accepted by smooth_current_boundary_3)")
        if n < 0:
            raise RuntimeError(f"smooth_pave(): not applicable to this episode (code {n}: see _capi.SMOOTH_*)")
        if not interior:
            st = self._vec.get_state(0)
            self.current_state = None if st["status"] & _capi.ST_NO_REFERENCE else self._vec.obs.cpu().numpy()[0].copy()
        return n

    def smooth(self, vertices=None, lr_1=0.999, lr_2=0.999, iteration=400):
        """MeshGeneration.smooth, general/mesh.py:1290-1392, on a finished episode (general/EBRD.py:391); `vertices` is
        implied (boundary.vertices).  Returns the number of sweeps; raises where the reference raises IndexError, and
        RuntimeError if the episode is still running (front > 5)."""
        from . import _capi
        sweeps, _ = self._vec.smooth(lr_1=lr_1, lr_2=lr_2, iteration=iteration)
        n = int(sweeps.cpu()[0])
        if n == _capi.SMOOTH_INDEX_ERROR:
            raise IndexError("list index out of range (smooth(): a vertex without a common neighbour, as in the reference)")
        if n < 0:
            raise RuntimeError(f"smooth(): not applicable to this episode (code {n}: see _capi.SMOOTH_*)")
        return n

    def extract_samples_2(self, meshes=None, n_neighbor=2, n_radius=3, radius=4, index=1, quality_threshold=0.7):
        """MeshGeneration.extract_samples_2, general/mesh.py:1438-1489, on this env's generated mesh (`meshes` is implied:
        the elements of the running episode), computed on the device (meshenv_extract_samples).  Returns the reference's
        three lists: all_samples (rows of 2 (2 n_neighbor + n_radius) floats), types ([1] / [0] / [0.5]), outputs."""
        samples, types, outputs, _, status = self._vec.extract_samples(n_neighbor, n_radius, float(radius), index, quality_threshold)
        if int(status.cpu()[0]) != 0:
            raise RuntimeError(f"extract_samples_2(): not applicable to this episode (status {int(status.cpu()[0])}: log overflow / "
                               "vertex degree / sector size, see include/meshenv.h)")
        return samples.cpu().tolist(), [[t] for t in types.cpu().tolist()], outputs.cpu().tolist()

    @property
    def not_valid_points(self):
        """[k, 2] coordinates of the reference vertices rejected since the last valid move (rl/boundary_env.py:47)."""
        return self._vec.get_not_valid(0)

    def seed(self, seed=None):
        if hasattr(self.action_space, "seed"):
            self.action_space.seed(seed)
        return [seed]

    def render(self, mode="human"):
        print(f"Generated elements: {len(self.generated_meshes)}")

    def close(self):
        self._vec.close()
