"""csn_amd — MI355X-native cross-shape attention (the hot path of marios2019/CSN's MID-FC/csa_models.py).

    from csn_amd.csa_models import get_model, CrossShapeAt, MultiHeadAttention, ScaledDotProductAttention

The attention arithmetic lives in csn_amd/libcsn_hip.so (C ABI: include/csn_hip.h, sources: csn_amd/csrc).
"""
from ._lib import CsnError, build, lib, LIB_PATH  # noqa: F401


def set_math_mode(mode) -> None:
    """Process-wide default arithmetic of the contractions: 'fp32' (exact fp32 matrix cores), 'bf16x3' (the library's
    default: three bf16 products per fp32 product, inside the 1e-4 contract), 'bf16' / 'fp16' (one product; outside it).
    Per model: ``get_model(..., math=...)``; per block of calls: ``csn_amd.functional.math_mode``."""
    from . import functional as CF
    from . import _lib
    _lib.check(lib().csn_set_math_mode(CF.mode_id(mode)), "csn_set_math_mode")

__version__ = "0.1.0"
