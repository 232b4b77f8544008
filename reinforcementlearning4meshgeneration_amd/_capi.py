"""ctypes binding of libmeshenv_hip.so (include/meshenv.h).  Fails loudly when the library is missing:
there is no CPU fallback anywhere in this package."""
from __future__ import annotations

import ctypes as C
import os

from .build import LIB_PATH

OBS_DIM = 18
ACT_DIM = 3

ST_NO_REFERENCE = 1
ST_LOG_OVERFLOW = 2
MOVE_OK, MOVE_NONE, MOVE_RAISES, MOVE_NEEDS_SMOOTHING, MOVE_SMOOTH_RAISES = 0, 1, 2, 3, 4
SMOOTH_SKIPPED, SMOOTH_LOG_OVERFLOW, SMOOTH_DEGREE, SMOOTH_NOT_FINISHED, SMOOTH_INDEX_ERROR, SMOOTH_RAISES = -1, -2, -3, -4, -5, -6
SMOOTH_NONFINITE = -7
E_ARG, E_HIP, E_RANGE, E_STATE = -1, -2, -3, -4


class MeshEnvParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("neighbor_num", C.c_int32), ("radius_num", C.c_int32),
        ("fail_limit", C.c_int32), ("log_capacity", C.c_int32), ("reserved", C.c_int32),
        ("radius", C.c_double), ("max_ref_angle", C.c_double), ("key_lambda", C.c_double),
        ("min_degree", C.c_double), ("max_degree", C.c_double), ("same_point_eps", C.c_double),
        ("ray_length", C.c_double),
    ]


class MeshEnvError(RuntimeError):
    pass


_lib = None

# every symbol include/meshenv.h declares
EXPORTS = [
    "meshenv_default_params", "meshenv_abi_version", "meshenv_device_count", "meshenv_create", "meshenv_destroy",
    "meshenv_last_error", "meshenv_set_stream", "meshenv_num_envs", "meshenv_max_ring", "meshenv_group_size", "meshenv_reset",
    "meshenv_step", "meshenv_rollout", "meshenv_get_status", "meshenv_get_state", "meshenv_get_elements",
    "meshenv_counters", "meshenv_set_timing", "meshenv_kernel_times", "meshenv_selftest", "meshenv_set_packed_output",
    "meshenv_actor_create", "meshenv_actor_destroy", "meshenv_actor_set_stream", "meshenv_actor_load",
    "meshenv_actor_forward", "meshenv_actor_sample", "meshenv_get_last_episode", "meshenv_element_quality",
    "meshenv_reset_static", "meshenv_move", "meshenv_get_not_valid", "meshenv_step_kernel",
    "meshenv_create_random", "meshenv_get_domain", "meshenv_smooth", "meshenv_smooth_final", "meshenv_get_not_valid_ids", "meshenv_step_actor",
    "meshenv_libm_exact", "meshenv_create_random_density", "meshenv_density_rings",
    "meshenv_step_actor_multi", "meshenv_extract_samples", "meshenv_atan2_exact", "meshenv_quad_quality",
]


def load():
    """Load the HIP library (no GPU needed for loading; compute calls need one)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MESHENV_LIB", LIB_PATH)  # A/B builds of the same library
    if not os.path.exists(path) and path == LIB_PATH:
        try:  # a fresh checkout: compile the library in-tree (needs hipcc); never a substitute implementation
            from .build import build
            build(force=True)
        except Exception as exc:
            raise MeshEnvError(f"{LIB_PATH} is missing and could not be built ({exc}); this package has no CPU "
                               "fallback") from exc
    if not os.path.exists(path):
        raise MeshEnvError(
            f"{path} is missing: build it with `python -m reinforcementlearning4meshgeneration_amd.build` "
            "(hipcc, gfx950).  This package has no CPU fallback.")
    try:
        # torch ships its own copy of the HIP runtime; loading it first makes this library bind to the same one (two
        # runtimes in one process do not share the device: torch.cuda.is_available() turns False if ours initialises first)
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    vp, i32p, f64p, u8p, f32p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_void_p, C.c_void_p
    L.meshenv_default_params.argtypes = [C.POINTER(MeshEnvParams)]
    L.meshenv_default_params.restype = None
    L.meshenv_abi_version.restype = C.c_int
    L.meshenv_device_count.restype = C.c_int
    L.meshenv_create.argtypes = [C.c_int, C.c_int, i32p, f64p, f64p, C.c_int, i32p, C.POINTER(MeshEnvParams), vp,
                                 C.POINTER(vp)]
    L.meshenv_create.restype = C.c_int
    L.meshenv_create_random.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_double, C.POINTER(MeshEnvParams), vp,
                                        C.POINTER(vp)]
    L.meshenv_create_random.restype = C.c_int
    L.meshenv_create_random_density.argtypes = [C.c_int, C.c_int, C.c_uint64, vp, C.c_int, C.c_double, C.c_double,
                                                C.POINTER(MeshEnvParams), vp, C.POINTER(vp), vp]
    L.meshenv_create_random_density.restype = C.c_int
    L.meshenv_density_rings.argtypes = [C.c_int, C.c_int, vp, vp, C.c_int, vp, C.c_double, vp, vp, vp, C.c_int64]
    L.meshenv_density_rings.restype = C.c_int
    L.meshenv_get_domain.argtypes = [vp, C.c_int, vp, C.c_int, i32p, vp]
    L.meshenv_get_domain.restype = C.c_int
    L.meshenv_destroy.argtypes = [vp]
    L.meshenv_destroy.restype = None
    L.meshenv_last_error.argtypes = [vp]
    L.meshenv_last_error.restype = C.c_char_p
    L.meshenv_set_stream.argtypes = [vp, vp]
    L.meshenv_num_envs.argtypes = [vp]
    L.meshenv_max_ring.argtypes = [vp]
    L.meshenv_group_size.argtypes = [vp]
    L.meshenv_group_size.restype = C.c_int
    L.meshenv_step_kernel.argtypes = [vp]
    L.meshenv_step_kernel.restype = C.c_int
    L.meshenv_libm_exact.argtypes = [vp]
    L.meshenv_libm_exact.restype = C.c_int
    L.meshenv_atan2_exact.argtypes = []
    L.meshenv_atan2_exact.restype = C.c_int
    L.meshenv_reset.argtypes = [vp, u8p, f32p]
    L.meshenv_reset_static.argtypes = [vp, u8p, f32p, C.c_int]
    L.meshenv_move.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.meshenv_get_not_valid.argtypes = [vp, C.c_int, vp, C.c_int, i32p]
    L.meshenv_get_not_valid_ids.argtypes = [vp, C.c_int, vp, C.c_int, i32p, vp]
    L.meshenv_get_not_valid_ids.restype = C.c_int
    L.meshenv_step_actor.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_uint64, C.c_uint64, vp, vp]
    L.meshenv_step_actor.restype = C.c_int
    L.meshenv_step_actor_multi.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_uint64, C.c_uint64, vp]
    L.meshenv_step_actor_multi.restype = C.c_int
    L.meshenv_extract_samples.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, vp, vp, vp, vp, vp, vp]
    L.meshenv_extract_samples.restype = C.c_int
    L.meshenv_smooth.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.meshenv_smooth.restype = C.c_int
    L.meshenv_smooth_final.argtypes = [vp, C.c_int, vp, C.c_int, C.c_double, C.c_double, vp, vp]
    L.meshenv_smooth_final.restype = C.c_int
    L.meshenv_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int]
    L.meshenv_rollout.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, C.c_int]
    L.meshenv_get_status.argtypes = [vp, vp]
    L.meshenv_get_state.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.meshenv_get_elements.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int, i32p, i32p]
    L.meshenv_get_last_episode.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int, i32p, i32p, i32p, i32p]
    L.meshenv_element_quality.argtypes = [vp, C.c_int, vp, vp, vp]
    L.meshenv_quad_quality.argtypes = [vp, C.c_int, vp, C.c_int, vp]
    L.meshenv_quad_quality.restype = C.c_int
    L.meshenv_counters.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.meshenv_set_timing.argtypes = [vp, C.c_int]
    L.meshenv_kernel_times.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int32)]
    L.meshenv_selftest.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    L.meshenv_selftest.restype = C.c_int
    L.meshenv_set_packed_output.argtypes = [vp, vp]
    L.meshenv_set_packed_output.restype = C.c_int
    L.meshenv_actor_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.meshenv_actor_destroy.argtypes = [vp]
    L.meshenv_actor_destroy.restype = None
    L.meshenv_actor_set_stream.argtypes = [vp, vp]
    L.meshenv_actor_load.argtypes = [vp] + [vp] * 12
    L.meshenv_actor_forward.argtypes = [vp, C.c_int, vp, vp, vp]
    L.meshenv_actor_sample.argtypes = [vp, C.c_int, vp, C.c_uint64, C.c_uint64, vp, vp]
    for name in ("meshenv_actor_create", "meshenv_actor_set_stream", "meshenv_actor_load", "meshenv_actor_forward",
                 "meshenv_actor_sample"):
        getattr(L, name).restype = C.c_int
    for name in ("meshenv_reset_static", "meshenv_move", "meshenv_get_not_valid", "meshenv_set_stream", "meshenv_num_envs", "meshenv_max_ring", "meshenv_reset", "meshenv_step",
                 "meshenv_rollout", "meshenv_get_status", "meshenv_get_state", "meshenv_get_elements",
                 "meshenv_counters", "meshenv_set_timing", "meshenv_kernel_times", "meshenv_get_last_episode",
                 "meshenv_element_quality"):
        getattr(L, name).restype = C.c_int
    _lib = L
    return L


def default_params() -> MeshEnvParams:
    p = MeshEnvParams()
    load().meshenv_default_params(C.byref(p))
    return p


def check(handle, rc: int, what: str):
    if rc != 0:
        msg = load().meshenv_last_error(handle)
        raise MeshEnvError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
