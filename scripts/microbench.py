#!/usr/bin/env python3
"""Kernel-level timing / accuracy of the plain contractions in both math modes (development aid)."""
import math, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csn_amd import _lib, functional as CF

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2]

def main():
    _lib.build()
    L = _lib.lib()
    rng = np.random.default_rng(0)
    S, C, N, R = 32, 256, 10000, 768
    x = torch.from_numpy(rng.standard_normal((S, C, N)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((R, C)) / 16).astype(np.float32)).cuda()
    dout = torch.from_numpy(rng.standard_normal((S, R, N)).astype(np.float32)).cuda()
    ref_p = torch.einsum("rc,cn->rn", w.double().cpu(), x[0].double().cpu())
    ref_w = torch.einsum("srn,scn->rc", dout[:2].double().cpu(), x[:2].double().cpu())
    for mode in (0, 1):
        L.csn_set_math_mode(mode)
        out = CF.project(x, w)
        e1 = ((out[0].cpu().double() - ref_p).abs().max() / ref_p.abs().max()).item()
        dw = CF.project_wgrad(dout[:2].contiguous(), x[:2].contiguous())
        e2 = ((dw.cpu().double() - ref_w).abs().max() / ref_w.abs().max()).item()
        t1 = timeit(lambda: CF.project(x, w))
        t2 = timeit(lambda: CF.project_wgrad(dout, x))
        fl = 2 * S * R * C * N
        print(f"mode {mode}: project (KN) {t1:7.3f} ms {fl / t1 / 1e9:7.1f} TF/s err {e1:.2e} | wgrad (NK split-K) {t2:7.3f} ms "
              f"{fl / t2 / 1e9:7.1f} TF/s err {e2:.2e}", flush=True)
    L.csn_set_math_mode(0)

if __name__ == "__main__":
    main()
