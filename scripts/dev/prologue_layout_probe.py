#!/usr/bin/env python3
"""How long does the attention forward's operand block (256 channels x 128 queries, fp32) take to arrive, as a function of the map's
row pitch?  Stamp build (scripts/dev/build_variant.sh stf -DCSN_STAMPS=0), one block of 128 queries per evaluation:
    pitch 128 (the block is one contiguous 128 KB piece)   against   pitch 10240 (rows 40 KB apart, as in the config-3 maps)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("CSN_LIB_PATH", "csn_amd/libcsn_stf.so")
from csn_amd import _lib, functional as CF
L = _lib.lib(); RAW = L._handle
H, d, T = 1, 256, 128
D = 256
L.csn_set_math_mode(1)
for ld, nb in ((128, 1), (10240, 80)):
    NP = ld
    E = 6400 if nb == 1 else 80            # both: >= 6144 work-groups, so that the stamped range 4096..6143 exists
    S = 64 if nb == 1 else 4
    q = torch.randn((S, D, NP), device="cuda") * 0.25
    kv = torch.randn((S, 2 * D, nb * 1024), device="cuda").bfloat16()
    qs = torch.arange(E, device="cuda", dtype=torch.int32) % S
    att = torch.empty((E, D, NP), device="cuda"); lse = torch.empty((E, H, NP), device="cuda")
    for _ in range(3):
        _lib.check(L.csn_block_attn_fwd_f32(q.data_ptr(), kv.data_ptr(), kv.data_ptr() + 2 * D * nb * 1024, D * NP, 2 * D * nb * 1024, CF._ptr(qs), CF._ptr(qs),
                                            NP, CF._ptr(att), D * NP, None, CF._ptr(lse), E, H, d, T, nb, 128, 8.0, 0.0, 1, 1, nb * 1024, CF._stream()), "fwd")
    torch.cuda.synchronize()
    pb = np.zeros(2048 * 8 * 8, dtype=np.uint64)
    RAW.csn_debug_read_pro.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
    RAW.csn_debug_read_pro(pb.ctypes.data, pb.nbytes)
    pr = pb.reshape(2048, 8, 8).astype(np.int64)
    ok = (pr[..., 7] > pr[..., 0]).all(axis=1) & (pr[..., 0] > 0).all(axis=1)
    dp = np.diff(pr[ok], axis=-1)
    print(f"row pitch {ld:6d} points: request operand block {dp[..., 0].mean():7.0f}  wait + barrier {dp[..., 1].mean():7.0f}  pick {dp[..., 2].mean():6.0f}  "
          f"tiles {dp[..., 4].mean() + dp[..., 5].mean():6.0f}  prologue {(pr[ok][..., 7] - pr[ok][..., 0]).mean():7.0f} cycles   ({ok.sum()} work-groups)")
    del q, kv, att, lse
