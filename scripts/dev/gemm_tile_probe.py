#!/usr/bin/env python3
"""development aid: the projection GEMM (256 rows x 256 channels x 10000 points per slot) on the 256 x 256 tiles (one
512-thread work-group per CU) against the 128 x 128 tiles (256 threads, two work-groups per CU)."""
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


def main():
    _lib.build()
    L = _lib.lib()
    L.csn_set_math_mode(1)
    S, C, N, R = 128, 256, 10000, 256
    x = torch.randn((S, C, N), device="cuda")
    w = torch.randn((R, C), device="cuda") / 16
    for rep in range(2):
        for big in (1, 0):
            L.csn_dev_set(0, big)
            t = timeit(lambda: CF.project(x, w))
            print(f"big_tiles={big}: project {S} x ({R} x {C}) x {N}: {t:.3f} ms  {2 * S * R * C * N / t / 1e9:.1f} TF/s "
                  f"{(S * C * N * 4 + S * R * N * 4) / t / 1e9:.2f} TB/s algorithmic", flush=True)
    L.csn_dev_set(0, 1)


if __name__ == "__main__":
    main()
