#!/usr/bin/env python3
"""csn_outproj_ln_bwd_f32 at config-3 size (256 evaluations x 256 ch x 10000 points, bf16x3): the fused LayerNorm-backward + dCtx
kernel (wx_lnb.hip) with its timing-only ablations against the two launches; the W_fc gradient is part of every call."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from csn_amd import _lib
_lib.build(); L = _lib.lib(); _lib.check(L.csn_set_math_mode(1))
E, C, D, NP, grp = 256, 256, 256, 10000, 8
g = torch.Generator(device="cuda").manual_seed(1)
dfeats = torch.randn((E // grp, C, NP), device="cuda", generator=g)
scale, rows = torch.randn((E, C), device="cuda", generator=g), torch.randn((E, C), device="cuda", generator=g)
xhat, rstd, ctx = torch.randn((E, C, NP), device="cuda", generator=g), torch.rand((E, NP), device="cuda", generator=g) + 0.5, torch.randn((E, D, NP), device="cuda", generator=g)
wfc_t = torch.randn((D, C), device="cuda", generator=g) / 16
dz, dctx, dw = torch.empty((E, C, NP), device="cuda"), torch.empty((E, D, NP), device="cuda"), torch.empty((C, D), device="cuda")
ws_n = L.csn_wgrad_workspace_floats(C, D, E, NP); ws = torch.empty((ws_n,), device="cuda")
st = torch.cuda.current_stream().cuda_stream
def call():
    _lib.check(L.csn_outproj_ln_bwd_f32(dfeats.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), C * NP, ctx.data_ptr(), D * NP, wfc_t.data_ptr(), dz.data_ptr(), None,
                                        dctx.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws_n, E, C, D, NP, NP, 0, 0.1, 77, 0, 0, rows.data_ptr(), E - 32, scale.data_ptr(), grp, st))
variants = [("two launches", 1), ("fused", 9), ("fused, no matrix instructions", 9 | 16), ("fused, no stores", 9 | 32), ("fused, no loads", 9 | 64),
            ("fused, no loads no stores", 9 | 96), ("fused, skeleton", 9 | 112)]
ts = {n: [] for n, _ in variants}
for rep in range(9):
    for n, v in variants:
        L.csn_dev_set(_lib.DEV_WX, v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize()
        if rep >= 2: ts[n].append(e0.elapsed_time(e1))
L.csn_dev_set(_lib.DEV_WX, _lib.DEV_WX_DEFAULT)
for n, _ in variants:
    print(f"{n:36s} {np.median(ts[n]):7.3f} ms  (the call: LayerNorm backward + dCtx + W_fc gradient)")
