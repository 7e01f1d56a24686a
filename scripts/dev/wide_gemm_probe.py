#!/usr/bin/env python3
"""development aid: the plain 256 x 256 projection products on the 8-wave kernel and on the experimental 16-wave kernel
(csn_debug_set_wide_gemm): same results, time side by side."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from csn_amd import _lib, functional as CF

def timeit(fn, n=7):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2]

L = _lib.lib()
rng = np.random.default_rng(0)
S, C, N, R = 128, 256, 10000, 256
x = torch.from_numpy(rng.standard_normal((S, C, N)).astype(np.float32)).cuda()
w = torch.from_numpy((rng.standard_normal((R, C)) / 16).astype(np.float32)).cuda()
x16 = x.bfloat16()
st = torch.cuda.current_stream().cuda_stream
for mode, name in ((1, "bf16x3"), (2, "bf16")):
    L.csn_set_math_mode(mode)
    cases = [("fp32 in, fp32 out", lambda: CF.project(x, w))]
    if mode == 2:
        out16 = torch.empty((S, R, N), device="cuda", dtype=torch.bfloat16)
        cases.append(("fp32 in, bf16 out", lambda: (_lib.check(L.csn_project_f32(x.data_ptr(), C * N, N, w.data_ptr(), R, C, out16.data_ptr(), R * N, N, S, N, 0, 1.0, 3, 0, st)), out16)[1]))
        cases.append(("bf16 in, fp32 out", lambda: CF.project(x16, w)))
    for label, fn in cases:
        res, t = {}, {}
        for rep in range(2):
            for wide in (0, 2):
                L.csn_dev_set(1, wide)
                res[wide] = fn().clone()
                t.setdefault(wide, []).append(timeit(fn))
        L.csn_dev_set(1, 1)
        same = torch.equal(res[0], res[2])
        d = (res[0].float() - res[1].float()).abs().max().item()
        print(f"{name:7s} {label}: 8 waves {min(t[0]):6.3f} ms   16 waves {min(t[2]):6.3f} ms   equal {same} (max diff {d:.2e})", flush=True)
L.csn_set_math_mode(1)
