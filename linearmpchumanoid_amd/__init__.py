"""MI355X-native batched NAO whole-body controller (LIPM-MPC + WBC QP + rigid-body terms).

Drop-in for the per-tick hot path of Ema158/linearMpcHumanoid behind the C ABI in include/lmh.h.
Everything numeric runs in hand-written gfx950 HIP kernels (linearmpchumanoid_amd/csrc).
"""
from . import capi  # noqa: F401
from .capi import LmhConfig, LmhError  # noqa: F401

__all__ = ["capi", "LmhConfig", "LmhError", "BatchedController", "default_config", "nominal_links"]


def __getattr__(name):
    # torch-dependent host layer is imported lazily so that ABI checks work without a GPU
    if name in ("BatchedController", "default_config", "nominal_links", "unpack_debug"):
        from . import controller
        return getattr(controller, name)
    raise AttributeError(name)
