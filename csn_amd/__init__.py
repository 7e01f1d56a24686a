"""csn_amd — MI355X-native cross-shape attention (the hot path of marios2019/CSN's MID-FC/csa_models.py).

    from csn_amd.csa_models import get_model, CrossShapeAt, MultiHeadAttention, ScaledDotProductAttention

The attention arithmetic lives in csn_amd/libcsn_hip.so (C ABI: include/csn_hip.h, sources: csn_amd/csrc).
"""
from ._lib import CsnError, build, lib, LIB_PATH  # noqa: F401

__version__ = "0.1.0"
