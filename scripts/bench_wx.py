#!/usr/bin/env python3
"""The K = 256 weight products of the config-3 step (bf16x3) on the streaming kernel (wx_stream.hip) and on the tiled GEMM kernels
they replace, interleaved in one process: ms per launch and the byte rate of each product's own bytes (read x once, write out)."""
import argparse
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csn_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--slots", type=int, default=128)
ap.add_argument("--evals", type=int, default=256)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--ld", type=int, default=10000, help="row pitch of the fp32 maps (>= 10000 points): 10016 makes every row start on a 128-byte line")
ap.add_argument("--ln", action="store_true", help="time the out-projection + LayerNorm instead")
a = ap.parse_args()
_lib.build()
L = _lib.lib()
_lib.check(L.csn_set_math_mode(1))
C, NP, T, nb = 256, 10000, 500, 20
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(1)
LD = a.ld
x = torch.randn((a.evals, C, LD), device="cuda", generator=g)
w = torch.randn((768, C), device="cuda", generator=g) / 16
ldp = nb * 1024
out_q = torch.empty((a.evals, 256, LD), device="cuda")
out_kv = torch.empty((a.slots, 512, ldp), device="cuda", dtype=torch.bfloat16)


def q_proj(n):
    _lib.check(L.csn_project_f32(x.data_ptr(), C * LD, LD, w.data_ptr(), 256, C, out_q.data_ptr(), 256 * LD, LD, n, NP, 256, 16.0, 0, 0, st))


def kv_proj(n):
    _lib.check(L.csn_project_f32(x.data_ptr(), C * LD, LD, w[256:].data_ptr(), 512, C, out_kv.data_ptr(), 512 * ldp, ldp, n, NP, 0, 1.0, 2, T, st))


def q_proj_blocked(n):
    # the same bytes as a BLOCKED map [point block of 32][256 channels][32 points]: every chunk is 32 KB contiguous
    nblk = n * (NP // 32)
    _lib.check(L.csn_project_f32(x.data_ptr(), C * 32, 32, w.data_ptr(), 256, C, out_q.data_ptr(), 256 * 32, 32, nblk, 32, 256, 16.0, 0, 0, st))


cases = [("Q projection, BLOCKED layout (128 slots x 312 blocks of 32 points)", lambda: q_proj_blocked(a.slots), a.slots * (NP // 32) * 32 * (C + 256) * 4),
         ("dCtx-shaped, BLOCKED layout (256 evaluations)", lambda: q_proj_blocked(a.evals), a.evals * (NP // 32) * 32 * (C + 256) * 4),
         ("Q projection (128 slots, fp32 out)", lambda: q_proj(a.slots), a.slots * (C + 256) * NP * 4),
         ("K/V projection (128 slots, tile planes out)", lambda: kv_proj(a.slots), a.slots * (C * NP * 4 + 512 * ldp * 2)),
         ("dCtx-shaped product (256 evaluations, fp32 out)", lambda: q_proj(a.evals), a.evals * (C + 256) * NP * 4)]
if a.ln:
    # out-projection + residual + LayerNorm + pooled sums (csn_outproj_ln_fwd_f32): 256 x 256 tiles (CSN_DEV_WX = 5) against the stream
    E = a.evals
    xres = torch.randn((a.slots, C, LD), device="cuda", generator=g)
    rid = (torch.arange(E, device="cuda", dtype=torch.int32) % a.slots).contiguous()
    xhat, rstd, sums = torch.empty((E, C, LD), device="cuda"), torch.empty((E, NP), device="cuda"), torch.empty((E, C), device="cuda")
    ws_n = L.csn_outproj_ln_workspace_floats(E, C, C, NP)
    ws = torch.empty((ws_n,), device="cuda")
    nbytes = E * C * NP * 4 * 3 + E * NP * 4
    for p_fc in (0.0, 0.1):
        times = {5: [], 1: []}
        for rep in range(a.reps + 2):
            for wx in (5, 1):
                L.csn_dev_set(_lib.DEV_WX, wx)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(L.csn_outproj_ln_fwd_f32(x.data_ptr(), C * LD, w.data_ptr(), xres.data_ptr(), C * LD, rid.data_ptr(), xhat.data_ptr(),
                                                    C * LD, rstd.data_ptr(), E, C, C, LD, NP, 1e-6, p_fc, 77, sums.data_ptr(), ws.data_ptr(), ws_n, st))
                e1.record()
                torch.cuda.synchronize()
                if rep >= 2:
                    times[wx].append(e0.elapsed_time(e1))
        L.csn_dev_set(_lib.DEV_WX, 1)
        t0, t1 = float(np.median(times[5])), float(np.median(times[1]))
        print(f"out-projection + LayerNorm + sums ({E} evaluations, fc dropout {p_fc}): tiled {t0:.3f} ms ({nbytes / t0 / 1e9:.2f} TB/s of its own bytes)   "
              f"streaming {t1:.3f} ms ({nbytes / t1 / 1e9:.2f} TB/s)   min {min(times[5]):.3f} / {min(times[1]):.3f}", flush=True)
    sys.exit(0)
ap2 = os.environ.get("WX_ABLATE")
if ap2:
    # timing-only ablations of the streaming kernel (outputs are wrong): CSN_DEV_WX value = 1 | 2 (lock step) | bits << 4
    for name, fn, nbytes in cases:
        row = []
        for label, v in (("full", 1), ("full staggered", 3), ("no mfma", 1 | 16), ("no stores", 1 | 32), ("no loads", 1 | 64), ("no mfma no stores", 1 | 48),
                         ("no mfma no loads", 1 | 80), ("skeleton", 1 | 112)):
            L.csn_dev_set(_lib.DEV_WX, v)
            ts = []
            for rep in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); torch.cuda.synchronize()
                if rep >= 2:
                    ts.append(e0.elapsed_time(e1))
            row.append(f"{label} {float(np.median(ts)):.3f}")
        L.csn_dev_set(_lib.DEV_WX, 1)
        print(name + ":  " + "   ".join(row), flush=True)
    sys.exit(0)
for name, fn, nbytes in cases:
    times = {0: [], 1: [], 3: []}
    for rep in range(a.reps + 2):
        for wx in (0, 3, 1):
            L.csn_dev_set(_lib.DEV_WX, wx)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            if rep >= 2:
                times[wx].append(e0.elapsed_time(e1))
    L.csn_dev_set(_lib.DEV_WX, 1)
    t0, t1, t3 = float(np.median(times[0])), float(np.median(times[1])), float(np.median(times[3]))
    print(f"{name}: tiled {t0:.3f} ms ({nbytes / t0 / 1e9:.2f} TB/s of its own bytes)   streaming, staggered {t3:.3f} ms ({nbytes / t3 / 1e9:.2f} TB/s)   "
          f"streaming {t1:.3f} ms ({nbytes / t1 / 1e9:.2f} TB/s)   min {min(times[0]):.3f} / {min(times[3]):.3f} / {min(times[1]):.3f}", flush=True)
