"""cheetah_amd -- MI355X (gfx950) batched Cheetah curve engine.

Host-side mirror of the toposware/cheetah API for the batched hot path; all
arithmetic runs in the HIP library behind include/cheetah_hip.h.
"""
from ._capi import AFFINE, JACOBIAN, PROJECTIVE, CheetahError  # noqa: F401


def __getattr__(name):  # lazy: importing the package must not need a GPU
    if name in ("Engine", "Fp6", "AffinePoint", "ProjectivePoint", "JacobianPoint", "BASEPOINT_TABLE", "BASEPOINT_LOOKUP", "BasePointTable", "default_engine"):
        from . import engine
        return getattr(engine, name)
    raise AttributeError(name)
