"""MI355X-native vectorised BoudaryEnv (quad element extraction) -- the step()/reset() hot path of
ZhuoQiuMcgill/ReinforcementLearning4MeshGeneration as hand-written HIP kernels behind a C-ABI.

    from reinforcementlearning4meshgeneration_amd import MeshVecEnv, SB3MeshVecEnv, BoudaryEnv, boundary, read_polygon
"""
from .domains import boundary, domain_constants, generate_polygon, random_domain, read_polygon  # noqa: F401

__all__ = ["MeshVecEnv", "SB3MeshVecEnv", "BoudaryEnv", "boundary", "read_polygon", "domain_constants", "generate_polygon",
           "random_domain", "MeshEnvError", "FusedActor"]


def __getattr__(name):  # torch / the HIP library are only needed once an environment is built
    if name == "MeshVecEnv":
        from .vec_env import MeshVecEnv
        return MeshVecEnv
    if name == "SB3MeshVecEnv":
        from .vec_env import SB3MeshVecEnv
        return SB3MeshVecEnv
    if name == "BoudaryEnv":
        from .boundary_env import BoudaryEnv
        return BoudaryEnv
    if name == "MeshEnvError":
        from ._capi import MeshEnvError
        return MeshEnvError
    if name == "FusedActor":
        from .actor import FusedActor
        return FusedActor
    raise AttributeError(name)
