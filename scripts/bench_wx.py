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
a = ap.parse_args()
_lib.build()
L = _lib.lib()
_lib.check(L.csn_set_math_mode(1))
C, NP, T, nb = 256, 10000, 500, 20
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn((a.evals, C, NP), device="cuda", generator=g)
w = torch.randn((768, C), device="cuda", generator=g) / 16
ldp = nb * 1024
out_q = torch.empty((a.evals, 256, NP), device="cuda")
out_kv = torch.empty((a.slots, 512, ldp), device="cuda", dtype=torch.bfloat16)


def q_proj(n):
    _lib.check(L.csn_project_f32(x.data_ptr(), C * NP, NP, w.data_ptr(), 256, C, out_q.data_ptr(), 256 * NP, NP, n, NP, 256, 16.0, 0, 0, st))


def kv_proj(n):
    _lib.check(L.csn_project_f32(x.data_ptr(), C * NP, NP, w[256:].data_ptr(), 512, C, out_kv.data_ptr(), 512 * ldp, ldp, n, NP, 0, 1.0, 2, T, st))


cases = [("Q projection (128 slots, fp32 out)", lambda: q_proj(a.slots), a.slots * (C + 256) * NP * 4),
         ("K/V projection (128 slots, tile planes out)", lambda: kv_proj(a.slots), a.slots * (C * NP * 4 + 512 * ldp * 2)),
         ("dCtx-shaped product (256 evaluations, fp32 out)", lambda: q_proj(a.evals), a.evals * (C + 256) * NP * 4)]
for name, fn, nbytes in cases:
    times = {0: [], 1: []}
    for rep in range(a.reps + 2):
        for wx in (0, 1):
            L.csn_dev_set(_lib.DEV_WX, wx)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            if rep >= 2:
                times[wx].append(e0.elapsed_time(e1))
    L.csn_dev_set(_lib.DEV_WX, 1)
    t0, t1 = float(np.median(times[0])), float(np.median(times[1]))
    print(f"{name}: tiled {t0:.3f} ms ({nbytes / t0 / 1e9:.2f} TB/s of its own bytes)   streaming {t1:.3f} ms ({nbytes / t1 / 1e9:.2f} TB/s)   "
          f"min {min(times[0]):.3f} / {min(times[1]):.3f}", flush=True)
