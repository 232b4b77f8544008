"""Fused SAC actor forward on the GPU (include/meshenv.h: meshenv_actor_*): the policy side of the rollout loop,
so that observation -> action -> step is two kernel launches with nothing leaving HBM.

Architecture = the reference's policy (rl/baselines/RL_Mesh.py:183-196: SB3 SAC, MlpPolicy, ReLU, [128, 128, 128]).
``FusedActor.from_torch(trunk, mu, log_std)`` takes the five ``torch.nn.Linear`` modules of an SB3 actor
(``actor.latent_pi`` layers 0/2/4, ``actor.mu``, ``actor.log_std``)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from .vec_env import ACTION_HIGH, ACTION_LOW


class FusedActor:
    def __init__(self, device: int = 0):
        import torch
        self._torch = torch
        self._L = _capi.load()
        if not torch.cuda.is_available():
            raise _capi.MeshEnvError("FusedActor needs a ROCm GPU")
        self.device = torch.device("cuda", device)
        self._h = C.c_void_p()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self._L.meshenv_actor_create(device, C.c_void_p(stream), C.byref(self._h))
        if rc != 0:
            raise _capi.MeshEnvError(f"meshenv_actor_create failed ({rc}): {self._L.meshenv_last_error(None).decode()}")
        self._stream = stream

    @classmethod
    def from_torch(cls, linears, mu, log_std, device: int = 0, low=ACTION_LOW, high=ACTION_HIGH):
        """linears: the three hidden torch.nn.Linear layers (18->128, 128->128, 128->128)."""
        self = cls(device)
        arrs = []
        for lin in list(linears) + [mu, log_std]:
            arrs.append(np.ascontiguousarray(lin.weight.detach().cpu().numpy(), np.float32))
            arrs.append(np.ascontiguousarray(lin.bias.detach().cpu().numpy(), np.float32))
        shapes = [a.shape for a in arrs]
        assert shapes == [(128, 18), (128,), (128, 128), (128,), (128, 128), (128,), (3, 128), (3,), (3, 128), (3,)], shapes
        arrs += [np.ascontiguousarray(low, np.float32), np.ascontiguousarray(high, np.float32)]
        rc = self._L.meshenv_actor_load(self._h, *[a.ctypes.data for a in arrs])
        if rc != 0:
            raise _capi.MeshEnvError(f"meshenv_actor_load failed ({rc})")
        return self

    def forward(self, obs, noise=None, out=None):
        """obs: float32 CUDA [n, 18]; noise: float32 CUDA [n, 3] of N(0,1) samples or None (deterministic).
        Returns actions float32 CUDA [n, 3] inside the action Box."""
        t = self._torch
        n = obs.shape[0]
        if out is None:
            out = t.empty((n, 3), dtype=t.float32, device=self.device)
        stream = t.cuda.current_stream(self.device).cuda_stream
        if stream != self._stream:
            self._L.meshenv_actor_set_stream(self._h, C.c_void_p(stream))
            self._stream = stream
        rc = self._L.meshenv_actor_forward(self._h, n, obs.data_ptr(), noise.data_ptr() if noise is not None else None,
                                           out.data_ptr())
        if rc != 0:
            raise _capi.MeshEnvError(f"meshenv_actor_forward failed ({rc})")
        return out

    def sample(self, obs, seed: int, counter: int, out=None, eps_out=None):
        """Stochastic action with the N(0,1) exploration noise drawn inside the kernel (Philox4x32-10 keyed by `seed`,
        counter (env, `counter`)): pass a fresh `counter` every rollout step.  eps_out (float32 CUDA [n,3], optional)
        receives the noise used, so that forward(obs, eps_out) reproduces the actions exactly."""
        t = self._torch
        n = obs.shape[0]
        if out is None:
            out = t.empty((n, 3), dtype=t.float32, device=self.device)
        stream = t.cuda.current_stream(self.device).cuda_stream
        if stream != self._stream:
            self._L.meshenv_actor_set_stream(self._h, C.c_void_p(stream))
            self._stream = stream
        rc = self._L.meshenv_actor_sample(self._h, n, obs.data_ptr(), C.c_uint64(seed & (2 ** 64 - 1)),
                                          C.c_uint64(counter & (2 ** 64 - 1)), out.data_ptr(),
                                          eps_out.data_ptr() if eps_out is not None else None)
        if rc != 0:
            raise _capi.MeshEnvError(f"meshenv_actor_sample failed ({rc})")
        return out

    def close(self):
        if self._h:
            self._L.meshenv_actor_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
