"""Vectorised BoudaryEnv on MI355X: thousands of independent boundary polygons advanced by one HIP kernel
launch per step (libmeshenv_hip.so), observations/actions/rewards as PyTorch-ROCm tensors.

Two call surfaces over the same engine:

* tensor-native -- ``reset() / step(actions) / rollout(actions)`` take and return CUDA tensors; nothing
  crosses PCIe.  This is what an on-device policy (SAC actor MLP) talks to.
* SB3 ``VecEnv`` -- ``SB3MeshVecEnv`` (below) IS a ``stable_baselines3.common.vec_env.VecEnv`` when SB3 is importable:
  ``reset() / step_async / step_wait / step / get_attr / env_method / envs[k] ...`` with numpy arrays, auto-reset,
  ``infos[k]["terminal_observation"]`` and ``infos[k]["TimeLimit.truncated"]``, i.e. the contract of the reference's
  vectorised caller (rl/baselines/dummy_vec_env.py:12-125), so ``SAC('MlpPolicy', SB3MeshVecEnv(...))`` passes SB3's
  ``isinstance(env, VecEnv)`` test and is not wrapped again.  ``MeshVecEnv`` itself offers the same numpy methods under
  ``reset_numpy / step_async / step_wait``; its ``reset()`` / ``step()`` are the tensor-native ones.

The Gym surface constants come from rl/boundary_env.py:27 (action Box) and :38-39 (observation Box).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _capi
from .compat import VEC_ENV_BASE
from .domains import Point, domain_constants
from .episode_tools import EpisodeTools

ACTION_LOW = np.array([-1.0, -1.5, 0.0], dtype=np.float32)   # rl/boundary_env.py:27
ACTION_HIGH = np.array([1.0, 1.5, 1.5], dtype=np.float32)
OBS_LOW, OBS_HIGH = -999.0, 999.0                            # rl/boundary_env.py:38-39
OBS_DIM = _capi.OBS_DIM


class Box:
    """Minimal stand-in for gym.spaces.Box, used only when neither gymnasium nor gym is importable."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is not None:
            self.low = np.full(tuple(shape), low, dtype=dtype)
            self.high = np.full(tuple(shape), high, dtype=dtype)
        else:
            self.low = np.asarray(low, dtype=dtype)
            self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"


def smoothing_log_capacity(ring_len: int, want: int = 4096) -> int:
    """Largest log_capacity <= want whose per-env smoothing graph still fits one CU's 160 KB of LDS
    (csrc/meshenv_smooth.h: smooth_lds_bytes = 16 B per ring slot + 60 B per logged vertex + 64, front smoother
    smooth_front_lds_bytes = 49 B per (ring slot + logged vertex) + 2 B per ring slot + 128, smooth_final_lds_bytes = 51 B per
    (ring slot + logged vertex) + 2 B per ring slot + 64; ring stride = ring length rounded up to 16).
    tests/test_host_cpu.py checks these formulas against the header."""
    cap = (int(ring_len) + 15) // 16 * 16
    lds = 160 * 1024
    by_interior = (lds - 64 - 16 * cap) // 60
    by_front = (lds - 128 - 2 * cap) // 49 - cap
    by_final = (lds - 64 - 2 * cap) // 51 - cap
    return int(max(16, min(want, by_interior, by_front, by_final, 65535 - cap)))


def make_spaces():
    """(observation_space, action_space) with the reference's bounds; real gymnasium/gym Boxes if available."""
    box = None
    for mod in ("gymnasium", "gym"):
        try:
            box = __import__(mod).spaces.Box
            break
        except Exception:
            continue
    if box is None:
        return Box(OBS_LOW, OBS_HIGH, shape=(OBS_DIM,), dtype=np.float32), Box(ACTION_LOW, ACTION_HIGH, dtype=np.float32)
    return (box(low=OBS_LOW, high=OBS_HIGH, shape=(OBS_DIM,), dtype=np.float32),
            box(low=ACTION_LOW, high=ACTION_HIGH, dtype=np.float32))


class MeshVecEnv:
    """n_envs boundary environments on one GPU.

    domains      list of polygons (each a clockwise list of (x, y)); see ``domains.py``
    env_domain   domain index per env (default: env k uses domain k % len(domains))
    device       CUDA/HIP device index
    log_capacity elements / new vertices logged per env per episode (needed for ``generated_meshes``)
    params       overrides of the geometry constants in MeshEnvParams (radius, max_ref_angle, key_lambda, min_degree,
                 max_degree, same_point_eps, ray_length); the reference's values are the default and the fast path
    """

    def __init__(self, domains: Sequence[Sequence[Point]], n_envs: Optional[int] = None,
                 env_domain: Optional[Sequence[int]] = None, device: int = 0, log_capacity: int = 0,
                 auto_reset: bool = True, fail_limit: int = 100, lazy_infos: bool = True,
                 params: Optional[dict] = None):
        import torch

        self._torch = torch
        self._L = _capi.load()
        if not torch.cuda.is_available():
            raise _capi.MeshEnvError("MeshVecEnv needs a ROCm GPU (torch.cuda.is_available() is False); "
                                     "this package has no CPU fallback")
        if len(domains) == 0:
            raise ValueError("at least one domain is required")
        if env_domain is None:
            n_envs = len(domains) if n_envs is None else int(n_envs)
            env_domain = np.arange(n_envs, dtype=np.int32) % len(domains)
        env_domain = np.ascontiguousarray(env_domain, dtype=np.int32)
        self.num_envs = int(len(env_domain))
        self.device = torch.device("cuda", device)
        self.auto_reset = bool(auto_reset)
        self.lazy_infos = bool(lazy_infos)
        self.log_capacity = int(log_capacity)
        self.domains = [list(d) for d in domains]
        self.constants = [domain_constants(d) for d in self.domains]
        self.env_domain = env_domain

        offs = np.zeros(len(domains) + 1, dtype=np.int32)
        for k, d in enumerate(self.domains):
            offs[k + 1] = offs[k] + len(d)
        xy = np.ascontiguousarray(np.concatenate([np.asarray(d, dtype=np.float64).reshape(-1, 2) for d in self.domains]))
        consts = np.ascontiguousarray(
            np.array([[c.original_area, c.est_min_l, c.est_crit_l] for c in self.constants], dtype=np.float64))
        prm = _capi.default_params()
        prm.log_capacity = self.log_capacity
        prm.fail_limit = int(fail_limit)
        for k, v in (params or {}).items():
            if k not in ("radius", "max_ref_angle", "key_lambda", "min_degree", "max_degree", "same_point_eps", "ray_length"):
                raise ValueError(f"unknown MeshEnvParams field {k!r}")
            setattr(prm, k, float(v))
        self._handle = C.c_void_p()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = self._L.meshenv_create(
                device, len(self.domains), offs.ctypes.data_as(C.POINTER(C.c_int32)),
                xy.ctypes.data_as(C.POINTER(C.c_double)), consts.ctypes.data_as(C.POINTER(C.c_double)),
                self.num_envs, env_domain.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(prm), C.c_void_p(stream),
                C.byref(self._handle))
        if rc != 0:
            msg = self._L.meshenv_last_error(None)
            raise _capi.MeshEnvError(f"meshenv_create failed (code {rc}): {msg.decode() if msg else ''}")
        self._finish_init()

    def _finish_init(self):
        torch = self._torch
        self.max_ring = self._L.meshenv_max_ring(self._handle)
        self.group_size = self._L.meshenv_group_size(self._handle)
        n = self.num_envs
        self.obs = torch.zeros((n, OBS_DIM), dtype=torch.float32, device=self.device)
        self.terminal_obs = torch.zeros((n, OBS_DIM), dtype=torch.float32, device=self.device)
        self.reward = torch.zeros(n, dtype=torch.float64, device=self.device)
        self.done = torch.zeros(n, dtype=torch.uint8, device=self.device)
        self.complete = torch.zeros(n, dtype=torch.uint8, device=self.device)
        self._status = torch.zeros(n, dtype=torch.uint8, device=self.device)
        self.observation_space, self.action_space = make_spaces()
        self._pending_actions = None
        self._closed = False
        self._envs = None
        if self._L.meshenv_atan2_exact() == 0:
            import warnings
            warnings.warn("meshenv: the running libm's atan2 could not be restated bit for bit (another glibc?): angles that "
                          "fall on a 1e-4 rounding boundary are decided by a correctly rounded atan2 instead -- results stay "
                          "within the 1e-5 tolerance of the reference but are no longer guaranteed bit-identical "
                          "(meshenv_atan2_exact() == 0)", RuntimeWarning, stacklevel=3)
        self.reset_tensor()

    @classmethod
    def from_random(cls, n_envs: int, seed: int, device: int = 0, num_verts: int = 0, edge: float = 0.45,
                    log_capacity: int = 0, auto_reset: bool = True, fail_limit: int = 100, lazy_infos: bool = True):
        """n_envs environments, env k on its own GenerateRandomPolygon-style ring of ``random.Random(seed + k)``
        (BASELINE.json configs[4]) -- generated, oriented, densified and measured ON THE DEVICE (meshenv_create_random);
        ``domains.random_domain(seed + k)`` is the same ring computed on the host.  ``self.domains`` / ``self.constants``
        are fetched from the device on first use."""
        import torch
        self = cls.__new__(cls)
        self._torch = torch
        self._L = _capi.load()
        if not torch.cuda.is_available():
            raise _capi.MeshEnvError("MeshVecEnv needs a ROCm GPU (torch.cuda.is_available() is False); "
                                     "this package has no CPU fallback")
        self.num_envs = int(n_envs)
        self.device = torch.device("cuda", device)
        self.auto_reset = bool(auto_reset)
        self.lazy_infos = bool(lazy_infos)
        self.log_capacity = int(log_capacity)
        self.env_domain = np.arange(self.num_envs, dtype=np.int32)
        self._domains = None
        self._constants = None
        prm = _capi.default_params()
        prm.log_capacity = self.log_capacity
        prm.fail_limit = int(fail_limit)
        self._handle = C.c_void_p()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = self._L.meshenv_create_random(device, self.num_envs, C.c_uint64(int(seed)), int(num_verts), float(edge),
                                               C.byref(prm), C.c_void_p(stream), C.byref(self._handle))
        if rc != 0:
            msg = self._L.meshenv_last_error(None)
            raise _capi.MeshEnvError(f"meshenv_create_random failed (code {rc}): {msg.decode() if msg else ''}")
        self._finish_init()
        return self

    @classmethod
    def from_random_density(cls, n_envs: int, seed: int, base_length: float = 45.0, density: float = 1.0, device: int = 0,
                            num_verts: int = 0, log_capacity: int = 0, auto_reset: bool = True, fail_limit: int = 100,
                            lazy_infos: bool = True, seeds: Optional[Sequence[int]] = None):
        """n_envs environments on GenerateRandomPolygon rings densified by the REFERENCE's Density.calculate_density
        (ui/tk-ui.py:252-276) on the device (meshenv_create_random_density); ``domains.random_density_domain(seed_k,
        base_length)`` is the same ring computed on the host.  calculate_density raises ZeroDivisionError for polygons with
        an edge of 0.5 - 1.5 spacings, so not every seed gives a domain: without ``seeds`` the seeds ``seed, seed + 1, ...``
        are probed on the device in blocks and the first n_envs that are defined are used (``self.seeds`` lists them);
        with ``seeds`` exactly those are used and an undefined one is an error."""
        import torch
        L = _capi.load()
        if not torch.cuda.is_available():
            raise _capi.MeshEnvError("MeshVecEnv needs a ROCm GPU (torch.cuda.is_available() is False); "
                                     "this package has no CPU fallback")
        n_envs = int(n_envs)
        dev = torch.device("cuda", device)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            if seeds is None:
                found, nxt = [], int(seed)
                while len(found) < n_envs:
                    blk = max(256, 2 * (n_envs - len(found)))
                    rz = np.zeros(blk, np.uint8)
                    rc = L.meshenv_create_random_density(device, blk, C.c_uint64(nxt), None, int(num_verts), float(base_length),
                                                         float(density), None, C.c_void_p(stream), None, rz.ctypes.data)
                    # E_STATE with flags set = some seeds of the block have no ring (flag 1: calculate_density raises, flag 2:
                    # not generated on the device): exactly those are left out; anything else is an error of the call itself
                    if rc not in (0, _capi.E_STATE) or (rc == _capi.E_STATE and not rz.any()):
                        msg = L.meshenv_last_error(None)
                        raise _capi.MeshEnvError(f"meshenv_create_random_density (probe) failed (code {rc}): {msg.decode() if msg else ''}")
                    found.extend(int(nxt + k) for k in np.nonzero(rz == 0)[0])
                    nxt += blk
                    if nxt - int(seed) > 256 * n_envs + 65536:
                        raise _capi.MeshEnvError("from_random_density: too few polygons on which calculate_density is defined "
                                                 f"at base_length {base_length}")
                seeds = found[:n_envs]
            seeds = np.ascontiguousarray(seeds, np.uint64)
            if seeds.shape != (n_envs,):
                raise ValueError(f"seeds must have {n_envs} entries")
            self = cls.__new__(cls)
            self._torch, self._L = torch, L
            self.num_envs, self.device = n_envs, dev
            self.auto_reset, self.lazy_infos, self.log_capacity = bool(auto_reset), bool(lazy_infos), int(log_capacity)
            self.env_domain = np.arange(n_envs, dtype=np.int32)
            self._domains = self._constants = None
            self.seeds = seeds.copy()
            prm = _capi.default_params()
            prm.log_capacity = self.log_capacity
            prm.fail_limit = int(fail_limit)
            self._handle = C.c_void_p()
            rz = np.zeros(n_envs, np.uint8)
            rc = L.meshenv_create_random_density(device, n_envs, C.c_uint64(0), seeds.ctypes.data, int(num_verts), float(base_length),
                                                 float(density), C.byref(prm), C.c_void_p(stream), C.byref(self._handle),
                                                 rz.ctypes.data)
        if rc != 0:
            msg = L.meshenv_last_error(None)
            raise _capi.MeshEnvError(f"meshenv_create_random_density failed (code {rc}): {msg.decode() if msg else ''}"
                                     + (f"; undefined seeds: {seeds[rz != 0][:8].tolist()}" if rz.any() else ""))
        self._finish_init()
        return self

    def get_domain(self, d: int):
        """(ring [n, 2] float64, (original_area, est_min_l ** 2, est_crit_l ** 2)) of domain d as the device holds it."""
        cap = self.max_ring
        xy = np.zeros(2 * cap, np.float64)
        n = C.c_int32(0)
        consts = np.zeros(3, np.float64)
        self._check(self._L.meshenv_get_domain(self._handle, int(d), xy.ctypes.data, cap, C.byref(n), consts.ctypes.data),
                    "meshenv_get_domain")
        return xy[:2 * n.value].reshape(-1, 2).copy(), tuple(float(c) for c in consts)

    @property
    def domains(self):
        if self._domains is None:   # from_random: fetched from the device on demand
            self._domains = [[tuple(p) for p in self.get_domain(d)[0]] for d in range(self.num_envs)]
        return self._domains

    @domains.setter
    def domains(self, value):
        self._domains = value

    @property
    def constants(self):
        if self._constants is None:
            self._constants = [domain_constants(d) for d in self.domains]
        return self._constants

    @constants.setter
    def constants(self, value):
        self._constants = value

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc, what):
        _capi.check(self._handle, rc, what)

    def _bind_stream(self):
        stream = self._torch.cuda.current_stream(self.device).cuda_stream
        if stream != getattr(self, "_stream", None):
            self._check(self._L.meshenv_set_stream(self._handle, C.c_void_p(stream)), "meshenv_set_stream")
            self._stream = stream

    def close(self):
        if not self._closed and self._handle:
            self._L.meshenv_destroy(self._handle)
            self._handle = C.c_void_p()
            self._closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ tensor-native API
    def reset_tensor(self, mask=None, static=False):
        """reset(static) of rl/boundary_env.py:67-84 for all envs (or those with mask != 0).  Returns obs [n,18].
        static=True is PointEnvironment(static=True): observation entry 1 carries 0 instead of the area ratio."""
        self._bind_stream()
        mptr = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=self._torch.uint8).contiguous()
            mptr = C.c_void_p(mask.data_ptr())
        self._check(self._L.meshenv_reset_static(self._handle, mptr, C.c_void_p(self.obs.data_ptr()), 1 if static else 0),
                    "meshenv_reset_static")
        return self.obs

    reset = reset_tensor      # SB3MeshVecEnv overrides reset() / step() with the VecEnv (numpy) forms

    def move(self, points, types):
        """move(new_point, type) of rl/boundary_env.py:265-432 for every env, one launch.
        points: float64 CUDA [n, 2] = (radius fraction, angle); types: float64 CUDA [n] rule selectors.
        Returns (obs [n,18] static form, done [n] uint8, complete [n] uint8, code [n] uint8 = _capi.MOVE_*); the reward is
        always 0 in the reference and is not returned.  No auto-reset: reset(mask=done | (code >= 2), static=True)."""
        t = self._torch
        points = points.to(device=self.device, dtype=t.float64).contiguous()
        types = types.to(device=self.device, dtype=t.float64).contiguous()
        if tuple(points.shape) != (self.num_envs, 2) or tuple(types.shape) != (self.num_envs,):
            raise ValueError(f"points must be [{self.num_envs}, 2] and types [{self.num_envs}]")
        if not hasattr(self, "move_code"):
            self.move_code = t.zeros(self.num_envs, dtype=t.uint8, device=self.device)
        self._bind_stream()
        rc = self._L.meshenv_move(self._handle, points.data_ptr(), types.data_ptr(), self.obs.data_ptr(),
                                  self.done.data_ptr(), self.complete.data_ptr(), self.move_code.data_ptr())
        self._check(rc, "meshenv_move")
        self._warn_if_libm_inexact()
        return self.obs, self.done, self.complete, self.move_code

    @property
    def step_kernel(self) -> str:
        """Name (as rocprofv3 prints it) of the kernel the next step() launches, from meshenv_step_kernel: the CU-group kernel
        for batches of one workgroup per CU, the one-wave-per-env kernel otherwise -- in its tie-breaking instantiation
        (`k_step<false, true, true>`) between a front smoothing and the next reset of all envs."""
        return {0: "meshenv::k_step<false, true, false, false, false>", 6: "meshenv::k_step<false, true, false, true, false>",
                7: "meshenv::k_step<false, true, false, false, true>", 8: "meshenv::k_step<false, true, false, true, true>", 1: f"meshenv::k_step_group<{self.group_size}, true, false, false>",
                2: f"meshenv::k_step_spec<{self.group_size}, true>",
                3: "meshenv::k_step<false, true, true, false, false>",
                4: "meshenv::k_step_group<16, true, true, false>",
                5: f"meshenv::k_step_group<{self.group_size}, true, false, true>"}[self._L.meshenv_step_kernel(self._handle)]

    @property
    def libm_exact(self) -> int:
        """meshenv_libm_exact: 1 = the smoothing kernels square like the running libm's pow(x, 2.0) (validated), 0 = they use
        x * x (another libm, or MESHENV_LIBM_EXACT=0), -1 = no smoothing call yet."""
        return int(self._L.meshenv_libm_exact(self._handle))

    @property
    def atan2_exact(self) -> int:
        """meshenv_atan2_exact: 1 = angles on a 1e-4 rounding boundary are decided by a validated restatement of the running
        libm's atan2 (so every quantised angle is the reference's), 0 = by a correctly rounded atan2 or ocml's."""
        return int(self._L.meshenv_atan2_exact())

    def _warn_if_libm_inexact(self):
        if not getattr(self, "_libm_warned", False) and self._L.meshenv_libm_exact(self._handle) == 0:
            import warnings
            self._libm_warned = True
            warnings.warn("meshenv: the smoothing kernels could not validate their restatement of the running libm's "
                          "pow(x, 2.0) (another glibc, or MESHENV_LIBM_EXACT=0): they square with x * x, which differs from "
                          "the reference by <= 1 ulp on distances -- smoothed coordinates are within tolerance but no longer "
                          "guaranteed bit-identical (meshenv_libm_exact() == 0)", RuntimeWarning, stacklevel=3)

    def smooth_pave(self, mask=None, iteration: int = 400, interior: bool = True, static: bool = False,
                    which: str = "current"):
        """MeshGeneration.smooth_pave(boundary.vertices, updated_boundary.vertices, iteration=iteration, interior=interior)
        (general/mesh.py:790-795) on the running episode of every env (mask: uint8/bool CUDA [n], None = all).  Needs
        log_capacity > 0; the vertex log (`generated_meshes`, the quality report) holds the moved coordinates afterwards.

        interior=True (general/EBRD.py:393): the generated vertices off the front are relaxed (smooth_fixed_vertices) and
        the candidate list is rebuilt; reference vertex and observation stay as they are, like in the reference.
        interior=False (what move() runs, rl/boundary_env.py:405-420): smooth_current_boundary_3 moves the front's
        generated vertices first; then, the front having moved, the point environment is recomputed at once
        (find_next_state(static=static)) and `self.obs` holds its observation.

        which="last" (interior=True only): the same relaxation on the archived episode of every env (what
        get_last_episode reads under auto-reset, e.g. an episode ended by truncation).

        Returns (sweeps int32 [n], diff float64 [n]); sweeps < 0: _capi.SMOOTH_* (SMOOTH_RAISES: the reference raises
        inside the front smoother; the env keeps what had moved until then)."""
        if which not in ("current", "last"):
            raise ValueError("which must be 'current' or 'last'")
        t = self._torch
        if not hasattr(self, "smooth_sweeps"):
            self.smooth_sweeps = t.zeros(self.num_envs, dtype=t.int32, device=self.device)
            self.smooth_diff = t.zeros(self.num_envs, dtype=t.float64, device=self.device)
        mptr = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=t.uint8).contiguous()
            if tuple(mask.shape) != (self.num_envs,):
                raise ValueError(f"mask must have shape ({self.num_envs},)")
            mptr = mask.data_ptr()
        self._bind_stream()
        rc = self._L.meshenv_smooth(self._handle, 1 if which == "last" else 0, mptr, int(iteration), 1 if interior else 0,
                                    1 if static else 0,
                                    self.smooth_sweeps.data_ptr(), self.smooth_diff.data_ptr(),
                                    None if interior else self.obs.data_ptr())
        self._check(rc, "meshenv_smooth")
        self._warn_if_libm_inexact()
        return self.smooth_sweeps, self.smooth_diff

    def smooth(self, mask=None, lr_1: float = 0.999, lr_2: float = 0.999, iteration: int = 400, which: str = "current"):
        """MeshGeneration.smooth(boundary.vertices, lr_1, lr_2, iteration) (general/mesh.py:1290-1392) on FINISHED
        meshes: which="current" -- running episodes that ended complete and were not reset yet (auto_reset=False);
        which="last" -- the archived episode of every env (what get_last_episode reads under auto-reset).  All generated
        vertices are relaxed, front included.  Returns (sweeps int32 [n], diff float64 [n]); sweeps < 0: _capi.SMOOTH_*
        (env untouched, e.g. SMOOTH_NOT_FINISHED for an episode that is still running / was truncated)."""
        if which not in ("current", "last"):
            raise ValueError("which must be 'current' or 'last'")
        t = self._torch
        if not hasattr(self, "smooth_sweeps"):
            self.smooth_sweeps = t.zeros(self.num_envs, dtype=t.int32, device=self.device)
            self.smooth_diff = t.zeros(self.num_envs, dtype=t.float64, device=self.device)
        mptr = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=t.uint8).contiguous()
            if tuple(mask.shape) != (self.num_envs,):
                raise ValueError(f"mask must have shape ({self.num_envs},)")
            mptr = mask.data_ptr()
        self._bind_stream()
        rc = self._L.meshenv_smooth_final(self._handle, 1 if which == "last" else 0, mptr, int(iteration), float(lr_1), float(lr_2),
                                          self.smooth_sweeps.data_ptr(), self.smooth_diff.data_ptr())
        self._check(rc, "meshenv_smooth_final")
        self._warn_if_libm_inexact()
        return self.smooth_sweeps, self.smooth_diff

    def get_not_valid(self, env: int) -> np.ndarray:
        """not_valid_points of one env (rl/boundary_env.py:47): [k, 2] coordinates of the reference vertices whose
        moves were rejected since the last valid move / reset."""
        cap = self.max_ring
        xy = np.zeros(2 * cap, np.float64)
        n = C.c_int32(0)
        self._check(self._L.meshenv_get_not_valid(self._handle, int(env), xy.ctypes.data, cap, C.byref(n)),
                    "meshenv_get_not_valid")
        return xy[:2 * n.value].reshape(-1, 2).copy()

    def get_not_valid_ids(self, env: int):
        """(ids of not_valid_points, (first id, last id, length, same_episode) of last_not_valid_points) of one env."""
        cap = self.max_ring
        ids = np.zeros(cap, np.int32)
        last = np.zeros(4, np.int32)
        n = C.c_int32(0)
        self._check(self._L.meshenv_get_not_valid_ids(self._handle, int(env), ids.ctypes.data, cap, C.byref(n),
                                                      last.ctypes.data), "meshenv_get_not_valid_ids")
        return ids[:n.value].copy(), tuple(int(x) for x in last)

    def step_tensor(self, actions):
        """One step() of every env.  actions: float32 CUDA tensor [n, 3].
        Returns (obs, reward, done, complete) -- views of buffers that the next call overwrites."""
        t = self._torch
        if actions.dtype != t.float32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=t.float32).contiguous()
        if actions.shape != (self.num_envs, 3):
            raise ValueError(f"actions must have shape ({self.num_envs}, 3), got {tuple(actions.shape)}")
        self._bind_stream()
        rc = self._L.meshenv_step(self._handle, actions.data_ptr(), self.obs.data_ptr(), self.reward.data_ptr(),
                                  self.done.data_ptr(), self.complete.data_ptr(), self.terminal_obs.data_ptr(),
                                  1 if self.auto_reset else 0)
        self._check(rc, "meshenv_step")
        return self.obs, self.reward, self.done, self.complete

    step = step_tensor

    def step_actor(self, actor, actions, seed: int = 0, counter: int = 0, sample: bool = True, eps_out=None):
        """One step() of every env AND the policy's actions for the next one, in one launch where the batch runs on the
        CU-group kernel (two launches otherwise; identical results): `actor` is a FusedActor on this device.
        Returns (obs, reward, done, complete, next_actions); next_actions is one of two internal ping-pong buffers, valid
        until the call after next -- pass it straight back as `actions`."""
        t = self._torch
        if actions.dtype != t.float32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=t.float32).contiguous()
        if actions.shape != (self.num_envs, 3):
            raise ValueError(f"actions must have shape ({self.num_envs}, 3), got {tuple(actions.shape)}")
        if not hasattr(self, "_act_pp"):
            self._act_pp = [t.empty((self.num_envs, 3), dtype=t.float32, device=self.device) for _ in range(2)]
        nxt = self._act_pp[0] if actions.data_ptr() != self._act_pp[0].data_ptr() else self._act_pp[1]
        self._bind_stream()
        stream = t.cuda.current_stream(self.device).cuda_stream
        if stream != actor._stream:
            actor._L.meshenv_actor_set_stream(actor._h, C.c_void_p(stream))
            actor._stream = stream
        rc = self._L.meshenv_step_actor(self._handle, actor._h, actions.data_ptr(), self.obs.data_ptr(), self.reward.data_ptr(),
                                        self.done.data_ptr(), self.complete.data_ptr(), self.terminal_obs.data_ptr(),
                                        1 if self.auto_reset else 0, 1 if sample else 0, C.c_uint64(seed & (2 ** 64 - 1)),
                                        C.c_uint64(counter & (2 ** 64 - 1)), nxt.data_ptr(),
                                        eps_out.data_ptr() if eps_out is not None else None)
        self._check(rc, "meshenv_step_actor")
        return self.obs, self.reward, self.done, self.complete, nxt

    def step_actor_T(self, actor, actions0, T: int, seed: int = 0, counter: int = 0, sample: bool = True,
                     want_terminal_obs: bool = False, want_eps: bool = False):
        """T vector steps of the closed loop (env step + policy) in ONE launch where the batch runs on the fused CU-group
        kernel (meshenv_step_actor_multi; T x step_actor otherwise, identical results).  actions0: [n, 3] actions of the first
        step.  Returns a dict of histories: actions [T+1, n, 3] (slice t + 1 = the policy's answer to the observations of
        step t; pass actions[T] as the next call's actions0), obs [T, n, 18], reward [T, n], done [T, n], complete [T, n],
        and terminal_obs [T, n, 18] / eps [T, n, 3] when asked for.  The noise counter of step t is counter + t."""
        t = self._torch
        n = self.num_envs
        T = int(T)
        acts = t.empty((T + 1, n, 3), dtype=t.float32, device=self.device)
        acts[0].copy_(actions0.to(device=self.device, dtype=t.float32).reshape(n, 3))
        out = dict(actions=acts, obs=t.empty((T, n, _capi.OBS_DIM), dtype=t.float32, device=self.device),
                   reward=t.empty((T, n), dtype=t.float64, device=self.device),
                   done=t.empty((T, n), dtype=t.uint8, device=self.device),
                   complete=t.empty((T, n), dtype=t.uint8, device=self.device))
        if want_terminal_obs:
            out["terminal_obs"] = t.zeros((T, n, _capi.OBS_DIM), dtype=t.float32, device=self.device)
        if want_eps:
            out["eps"] = t.empty((T, n, 3), dtype=t.float32, device=self.device)
        self._bind_stream()
        stream = t.cuda.current_stream(self.device).cuda_stream
        if stream != actor._stream:
            actor._L.meshenv_actor_set_stream(actor._h, C.c_void_p(stream))
            actor._stream = stream
        rc = self._L.meshenv_step_actor_multi(self._handle, actor._h, T, acts.data_ptr(), out["obs"].data_ptr(), out["reward"].data_ptr(),
                                          out["done"].data_ptr(), out["complete"].data_ptr(),
                                          out["terminal_obs"].data_ptr() if want_terminal_obs else None,
                                          1 if self.auto_reset else 0, 1 if sample else 0, C.c_uint64(seed & (2 ** 64 - 1)),
                                          C.c_uint64(counter & (2 ** 64 - 1)), out["eps"].data_ptr() if want_eps else None)
        self._check(rc, "meshenv_step_actor_multi")
        self.obs.copy_(out["obs"][T - 1])      # the env's current observation / flags, as after T single steps
        self.reward.copy_(out["reward"][T - 1]); self.done.copy_(out["done"][T - 1]); self.complete.copy_(out["complete"][T - 1])
        return out

    def extract_samples(self, n_neighbor: int = 2, n_radius: int = 3, radius: float = 4.0, index: int = 1,
                        quality_threshold: float = 0.7, which: str = "current", mask=None):
        """MeshGeneration.extract_samples_2 (general/mesh.py:1438-1489) for the generated mesh of every env, on the device
        (meshenv_extract_samples: a counting launch, a prefix sum, a filling launch).  Returns (samples [total, 2 (2 n_neighbor +
        n_radius)], types [total], outputs [total, 2], offsets [n + 1] int64, status [n] uint8) as CUDA tensors: env k's
        samples are rows offsets[k] .. offsets[k + 1], in the reference's order."""
        if which not in ("current", "last"):
            raise ValueError("which must be 'current' or 'last'")
        t = self._torch
        n = self.num_envs
        cnt = t.zeros(n, dtype=t.int64, device=self.device)
        st = t.zeros(n, dtype=t.uint8, device=self.device)
        mptr = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=t.uint8).contiguous()
            mptr = mask.data_ptr()
        self._bind_stream()
        w = 1 if which == "last" else 0
        args = (int(n_neighbor), int(n_radius), float(radius), int(index), float(quality_threshold))
        self._check(self._L.meshenv_extract_samples(self._handle, w, mptr, *args, cnt.data_ptr(), st.data_ptr(), None, None, None, None),
                    "meshenv_extract_samples")
        offs = t.zeros(n + 1, dtype=t.int64, device=self.device)
        offs[1:] = t.cumsum(cnt, 0)
        total = int(offs[-1].item())
        row = 2 * (2 * int(n_neighbor) + int(n_radius))
        samples = t.empty((max(total, 1), row), dtype=t.float64, device=self.device)
        outputs = t.empty((max(total, 1), 2), dtype=t.float64, device=self.device)
        types = t.empty(max(total, 1), dtype=t.float64, device=self.device)
        if total:
            self._check(self._L.meshenv_extract_samples(self._handle, w, mptr, *args, cnt.data_ptr(), st.data_ptr(), offs.data_ptr(),
                                                        samples.data_ptr(), outputs.data_ptr(), types.data_ptr()),
                        "meshenv_extract_samples")
        return samples[:total], types[:total], outputs[:total], offs, st

    def rollout(self, actions):
        """T consecutive steps in one kernel launch.  actions: float32 CUDA tensor [T, n, 3].
        Returns (obs_after_last_step [n,18], reward [T,n], done [T,n], complete [T,n])."""
        t = self._torch
        if actions.dtype != t.float32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=t.float32).contiguous()
        if actions.dim() != 3 or actions.shape[1:] != (self.num_envs, 3):
            raise ValueError(f"actions must have shape (T, {self.num_envs}, 3), got {tuple(actions.shape)}")
        T = int(actions.shape[0])
        reward = t.empty((T, self.num_envs), dtype=t.float64, device=self.device)
        done = t.empty((T, self.num_envs), dtype=t.uint8, device=self.device)
        complete = t.empty((T, self.num_envs), dtype=t.uint8, device=self.device)
        self._bind_stream()
        rc = self._L.meshenv_rollout(self._handle, T, actions.data_ptr(), self.obs.data_ptr(), reward.data_ptr(),
                                     done.data_ptr(), complete.data_ptr(), 1 if self.auto_reset else 0)
        self._check(rc, "meshenv_rollout")
        return self.obs, reward, done, complete

    def set_packed_output(self, msg):
        """msg: float32 CUDA tensor [n, 21] (or None) that every following step fills with
        (obs | reward | done | complete) -- the all_gather payload of the multi-GPU path."""
        self._packed_user = msg
        if msg is None:
            self._check(self._L.meshenv_set_packed_output(self._handle, None), "meshenv_set_packed_output")
            return
        t = self._torch
        if msg.dtype != t.float32 or tuple(msg.shape) != (self.num_envs, 21) or not msg.is_contiguous() or msg.device != self.device:
            raise ValueError("packed output must be a contiguous float32 CUDA tensor [n_envs, 21]")
        self._check(self._L.meshenv_set_packed_output(self._handle, C.c_void_p(msg.data_ptr())), "meshenv_set_packed_output")

    def status(self):
        self._bind_stream()
        self._check(self._L.meshenv_get_status(self._handle, self._status.data_ptr()), "meshenv_get_status")
        return self._status

    # ------------------------------------------------------------------ introspection (host side)
    def get_state(self, env: int) -> dict:
        m = self.max_ring
        ids = np.zeros(m, np.int32)
        xy = np.zeros(2 * m, np.float64)
        key = np.zeros(m, np.float64)
        stamp = np.zeros(m, np.int32)
        sc = np.zeros(8, np.int32)
        fs = np.zeros(2, np.float64)
        rc = self._L.meshenv_get_state(self._handle, int(env), ids.ctypes.data, xy.ctypes.data, key.ctypes.data,
                                       stamp.ctypes.data, sc.ctypes.data, fs.ctypes.data)
        self._check(rc, "meshenv_get_state")
        n = int(sc[0])
        ids, xy, key, stamp = ids[:n], xy[:2 * n].reshape(-1, 2), key[:n], stamp[:n]
        cand = np.nonzero(stamp != np.iinfo(np.int32).min)[0]
        order = sorted(cand.tolist(), key=lambda i: (key[i], -int(stamp[i])))
        return dict(n=n, ring_ids=ids, ring_xy=xy, cand_key=key, cand_stamp=stamp,
                    cand_order_ids=ids[order] if order else np.zeros(0, np.int32), cand_order_keys=key[order],
                    ref_index=int(sc[1]), ref_id=int(ids[sc[1]]) if sc[1] >= 0 else -1, n_elem=int(sc[2]),
                    failed_num=int(sc[3]), n_vert=int(sc[4]), status=int(sc[5]), domain=int(sc[6]), n0=int(sc[7]),
                    current_area=float(fs[0]), base_length=float(fs[1]))

    def get_elements(self, env: int):
        """(quads [n_elem,4] global vertex ids, vertex_xy [n_vert,2]) of the env's current episode."""
        if self.log_capacity <= 0:
            raise _capi.MeshEnvError("create the MeshVecEnv with log_capacity > 0 to read generated meshes")
        cap_e = self.log_capacity
        cap_v = self.max_ring + self.log_capacity
        quads = np.zeros(4 * cap_e, np.int32)
        vxy = np.zeros(2 * cap_v, np.float64)
        ne, nv = C.c_int32(0), C.c_int32(0)
        rc = self._L.meshenv_get_elements(self._handle, int(env), quads.ctypes.data, cap_e, vxy.ctypes.data, cap_v,
                                          C.byref(ne), C.byref(nv))
        self._check(rc, "meshenv_get_elements")
        if ne.value >= cap_e and int(self.status().cpu()[int(env)]) & _capi.ST_LOG_OVERFLOW:
            import warnings   # the log holds the first log_capacity entries only
            warnings.warn(f"env {env}: the episode outgrew log_capacity = {self.log_capacity}: the mesh returned (and any "
                          "export, quality report or smoothing of it) is truncated -- create the environment with a larger "
                          "log_capacity", RuntimeWarning, stacklevel=2)
        return quads[:4 * ne.value].reshape(-1, 4).copy(), vxy[:2 * nv.value].reshape(-1, 2).copy()

    def get_last_episode(self, env: int):
        """The last FINISHED episode of the env (kept across auto-reset): dict(quads [n_elem,4], vertex_xy [n_vert,2],
        is_complete, overflow, episodes).  episodes == 0: nothing finished yet."""
        if self.log_capacity <= 0:
            raise _capi.MeshEnvError("create the MeshVecEnv with log_capacity > 0 to read generated meshes")
        cap_e = self.log_capacity
        cap_v = self.max_ring + self.log_capacity
        quads = np.zeros(4 * cap_e, np.int32)
        vxy = np.zeros(2 * cap_v, np.float64)
        ne, nv, fl, ep = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
        rc = self._L.meshenv_get_last_episode(self._handle, int(env), quads.ctypes.data, cap_e, vxy.ctypes.data, cap_v,
                                              C.byref(ne), C.byref(nv), C.byref(fl), C.byref(ep))
        self._check(rc, "meshenv_get_last_episode")
        return dict(quads=quads[:4 * ne.value].reshape(-1, 4).copy(), vertex_xy=vxy[:2 * nv.value].reshape(-1, 2).copy(),
                    is_complete=bool(fl.value & 1), overflow=bool(fl.value & 2), episodes=int(ep.value))

    QUALITY_MEASURES = ("min_angle_deg", "max_angle_deg", "scaled_jacobian", "stretch", "taper", "robust", "area",
                        "default")

    def element_quality(self, which: str = "last", per_element: bool = True):
        """Quality report of every env's mesh in one launch (general/components.py:863-950 measures).

        which: "last" = the archived finished episodes, "current" = the running ones.
        Returns (records [n, log_capacity, 8] float64 or None, stats [n, 8, 4] = min/mean/max/variance, counts [n] int32),
        all CUDA tensors; records rows >= counts[k] are zero."""
        if self.log_capacity <= 0:
            raise _capi.MeshEnvError("create the MeshVecEnv with log_capacity > 0 to read generated meshes")
        torch = self._torch
        n = self.num_envs
        rec = torch.zeros((n, self.log_capacity, 8), dtype=torch.float64, device=self.device) if per_element else None
        stats = torch.empty((n, 8, 4), dtype=torch.float64, device=self.device)
        counts = torch.empty(n, dtype=torch.int32, device=self.device)
        rc = self._L.meshenv_element_quality(self._handle, {"current": 0, "last": 1}[which],
                                             rec.data_ptr() if per_element else None, stats.data_ptr(), counts.data_ptr())
        self._check(rc, "meshenv_element_quality")
        return rec, stats, counts

    def quad_quality(self, quad_xy, index: int = 0) -> np.ndarray:
        """MeshGeneration.get_quality(element, index) (general/mesh.py:1728-1747) of m arbitrary quads on the device
        (meshenv_quad_quality): quad_xy [m, 4, 2] (array or CUDA float64 tensor, Mesh.vertices order); index 0 'default',
        1 compute_element_quality, 3 'stretch', 4 'robust', 5 'strong'.  Returns a float64 array [m]."""
        t = self._torch
        q = t.as_tensor(np.asarray(quad_xy, np.float64) if not t.is_tensor(quad_xy) else quad_xy, dtype=t.float64,
                        device=self.device).reshape(-1, 4, 2).contiguous()
        out = t.empty(q.shape[0], dtype=t.float64, device=self.device)
        self._bind_stream()
        self._check(self._L.meshenv_quad_quality(self._handle, int(q.shape[0]), q.data_ptr(), int(index), out.data_ptr()),
                    "meshenv_quad_quality")
        return out.cpu().numpy()

    def quality_report(self, which: str = "last") -> dict:
        """Per measure: the mean over meshes of (average, standard deviation) -- what verdict() prints for a list of
        domains (Measurement/quality_verdict.py:93-105) -- plus the overall range; meshes without elements are skipped."""
        _, stats, counts = self.element_quality(which, per_element=False)
        stats = stats.cpu().numpy(); counts = counts.cpu().numpy()
        live = counts > 0
        out = {"meshes": int(live.sum()), "elements": int(counts.sum())}
        for k, name in enumerate(self.QUALITY_MEASURES):
            if live.any():
                st = stats[live, k]
                out[name] = dict(average=float(st[:, 1].mean()), std=float(np.sqrt(np.abs(st[:, 3])).mean()),
                                 min=float(st[:, 0].min()), max=float(st[:, 2].max()))
        return out

    def counters(self) -> dict:
        out = (C.c_uint64 * 4)()
        self._check(self._L.meshenv_counters(self._handle, out), "meshenv_counters")
        return dict(steps=int(out[0]), valid=int(out[1]), sum_ring=int(out[2]), sum_ring_valid=int(out[3]))

    def set_timing(self, every: int):
        """Bracket every other group of `every` consecutive step/rollout launches with HIP events (0 = off);
        kernel_times_ms() then returns the per-launch average of each bracketed group."""
        self._check(self._L.meshenv_set_timing(self._handle, int(every)), "meshenv_set_timing")

    def kernel_times_ms(self) -> np.ndarray:
        """Durations (ms) of the step/rollout launches recorded since set_timing(True) / the last call."""
        buf = np.zeros(4096, np.float32)
        n = C.c_int32(0)
        self._check(self._L.meshenv_kernel_times(self._handle, buf.ctypes.data, len(buf), C.byref(n)),
                    "meshenv_kernel_times")
        return buf[:n.value].copy()

    # ------------------------------------------------------------------ SB3 VecEnv-shaped API (numpy)
    def step_async(self, actions):
        self._pending_actions = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.num_envs, 3)

    def _numpy_buffers(self):
        """Pinned host staging for the numpy API: one upload (actions) and ONE download per step -- the step kernel writes
        the packed (obs | reward | done | complete) message (meshenv_set_packed_output) and that [n, 21] float32 block is
        what crosses PCIe, instead of four separate tensors."""
        if getattr(self, "_np_msg_dev", None) is None:
            t, n = self._torch, self.num_envs
            self._np_act_host = t.empty((n, 3), dtype=t.float32, pin_memory=True)
            self._np_act_dev = t.empty((n, 3), dtype=t.float32, device=self.device)
            self._np_msg_dev = t.zeros((n, 21), dtype=t.float32, device=self.device)
            self._np_msg_host = t.empty((n, 21), dtype=t.float32, pin_memory=True)
            self._np_term_host = t.empty((n, OBS_DIM), dtype=t.float32, pin_memory=True)
        return self._np_act_host, self._np_act_dev, self._np_msg_dev, self._np_msg_host

    def step_wait(self):
        t = self._torch
        if self._pending_actions is None:
            raise RuntimeError("step_wait() without step_async()")
        act_host, act_dev, msg_dev, msg_host = self._numpy_buffers()
        act_host.numpy()[:] = self._pending_actions
        self._pending_actions = None
        act_dev.copy_(act_host, non_blocking=True)
        user_msg = getattr(self, "_packed_user", None)   # a caller's own exchange buffer (multi-GPU path) is put back
        self.set_packed_output(msg_dev)
        try:
            self.step_tensor(act_dev)
        finally:
            self.set_packed_output(user_msg)
        msg_host.copy_(msg_dev, non_blocking=True)
        self._np_term_host.copy_(self.terminal_obs, non_blocking=True)   # rows of finished envs are read below
        t.cuda.current_stream(self.device).synchronize()
        m = msg_host.numpy()
        obs_np = m[:, :OBS_DIM].copy()
        rew_np = m[:, 18].copy()            # float32((double) reward): what .astype(np.float32) of the float64 reward gives
        done_np = m[:, 19] != 0
        comp_np = m[:, 20] != 0
        infos = self._build_infos(done_np, comp_np)
        return obs_np, rew_np, done_np, infos

    def step_numpy(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def _build_infos(self, done_np, comp_np) -> List[dict]:
        if self.lazy_infos:
            shared = {"is_complete": True}
            infos = [shared] * self.num_envs
        else:
            infos = [{"is_complete": bool(c)} for c in comp_np]
        idx = np.nonzero(done_np)[0]
        if idx.size:
            infos = list(infos)
            term = self._np_term_host.numpy()    # downloaded with the message, before the step's one synchronise
            for k in idx:
                infos[k] = {"is_complete": bool(comp_np[k]), "terminal_observation": term[k].copy(),
                            "TimeLimit.truncated": not bool(comp_np[k])}
        return infos

    def reset_numpy(self):
        return self.reset_tensor().cpu().numpy()

    def seed(self, seed=None):
        # step()/reset() of the reference contain no RNG; only the action space sampler is seeded
        if hasattr(self.action_space, "seed"):
            self.action_space.seed(seed)
        return [seed] * self.num_envs

    @property
    def envs(self):
        """Per-env views, indexable like DummyVecEnv.envs (rl/baselines/dummy_vec_env.py:25): ``env.envs[0].generated_meshes``,
        ``env.envs[0].save_meshes(...)`` of the evaluation callback (rl/baselines/CustomizeCallback.py:131-133)."""
        if self._envs is None:
            self._envs = _EnvViews(self)
        return self._envs

    def _indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        return [indices] if isinstance(indices, (int, np.integer)) else list(indices)

    def get_attr(self, attr_name, indices=None):
        return [getattr(self.envs[i], attr_name) for i in self._indices(indices)]

    def set_attr(self, attr_name, value, indices=None):
        raise AttributeError(f"MeshVecEnv has no settable per-env attribute {attr_name!r}")

    def env_method(self, method_name, *args, indices=None, **kwargs):
        return [getattr(self.envs[i], method_name)(*args, **kwargs) for i in self._indices(indices)]

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False for _ in self._indices(indices)]

    def render(self, mode="human"):
        return None


class EnvView(EpisodeTools):
    """Environment k of a MeshVecEnv as the reference's callers address one env: ``generated_meshes``, ``save_meshes``,
    ``get_quality``, the exports (EpisodeTools) and the state fields of ``MeshVecEnv.get_state`` as attributes."""

    render_mode = None          # what SB3 2.x's VecEnv.__init__ asks every env for

    def __init__(self, vec: MeshVecEnv, k: int):
        self._vec, self._k = vec, int(k)

    @property
    def points(self):
        return self._vec.domains[int(self._vec.env_domain[self._k])]

    @property
    def observation_space(self):
        return self._vec.observation_space

    @property
    def action_space(self):
        return self._vec.action_space

    def get_last_episode(self):
        return self._vec.get_last_episode(self._k)

    def __getattr__(self, name):            # n, ring_ids, ring_xy, ref_id, n_elem, status, ...
        if name.startswith("_"):
            raise AttributeError(name)
        st = self._vec.get_state(self._k)
        if name in st:
            return st[name]
        raise AttributeError(name)


class _EnvViews:
    """Lazy sequence of EnvView (a 65 536-env batch does not build 65 536 objects to serve ``envs[0]``)."""

    def __init__(self, vec):
        self._vec, self._made = vec, {}

    def __len__(self):
        return self._vec.num_envs

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        k = int(k)
        if k < 0:
            k += len(self)
        if not 0 <= k < len(self):
            raise IndexError(k)
        if k not in self._made:
            self._made[k] = EnvView(self._vec, k)
        return self._made[k]

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class SB3MeshVecEnv(MeshVecEnv, VEC_ENV_BASE):
    """The vectorised drop-in for Stable-Baselines3: a ``stable_baselines3.common.vec_env.VecEnv`` subclass (when SB3 is
    importable; a plain class with the same methods otherwise) over the HIP engine, replacing ``DummyVecEnv([lambda: env])``
    of rl/baselines/dummy_vec_env.py:12-125 -- ``SAC('MlpPolicy', SB3MeshVecEnv([boundary(0)], n_envs=4096), ...)``.

    ``reset()`` -> obs float32 [n, 18] (numpy); ``step_async(actions)`` / ``step_wait()`` / ``step(actions)`` ->
    (obs, float32 rewards, bool dones, infos) with auto-reset, ``infos[k]["terminal_observation"]`` and
    ``infos[k]["TimeLimit.truncated"]`` on the envs that finished; ``envs[k]`` / ``get_attr`` / ``env_method`` reach the
    per-env tools (``generated_meshes``, ``save_meshes``, ...).  The tensor-native calls stay available as
    ``reset_tensor()`` / ``step_tensor()``.  auto_reset=False gives the reference's own DummyVecEnv variant, whose reset is
    commented out (rl/baselines/dummy_vec_env.py:49) for the evaluation callback."""

    def _finish_init(self):      # (also reached from the from_random / from_random_density constructors)
        self.lazy_infos = False  # SB3 (VecMonitor, callbacks) writes into infos[k]: one dict per env, never a shared one
        MeshVecEnv._finish_init(self)
        if VEC_ENV_BASE is not object:
            n = self.num_envs            # VecEnv.__init__ assigns num_envs / observation_space / action_space (+ SB3 2.x bookkeeping)
            VEC_ENV_BASE.__init__(self, n, self.observation_space, self.action_space)
        if not hasattr(self, "reset_infos"):
            self.reset_infos = [{} for _ in range(self.num_envs)]

    def reset(self):
        obs = self.reset_numpy()
        self.reset_infos = [{} for _ in range(self.num_envs)]
        if hasattr(self, "_reset_seeds"):
            self._reset_seeds()
        if hasattr(self, "_reset_options"):
            self._reset_options()
        return obs

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        MeshVecEnv.close(self)

    def get_images(self):
        return [None for _ in range(self.num_envs)]

    def render(self, mode=None):
        return None
