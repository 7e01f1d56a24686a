#!/usr/bin/env python3
"""development aid: the same projection GEMM on random and on all-zero operands (power / clock sensitivity)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csn_amd import _lib, functional as CF
from microbench import timeit
L = _lib.lib(); L.csn_set_math_mode(1)
S, C, N, R = 32, 256, 10000, 768
for name, x, w in (("random", torch.randn((S, C, N), device="cuda"), torch.randn((R, C), device="cuda") / 16),
                   ("zeros", torch.zeros((S, C, N), device="cuda"), torch.zeros((R, C), device="cuda"))):
    for big in (1, 0):
        L.csn_dev_set(0, big)
        t = timeit(lambda: CF.project(x, w), n=9)
        print(f"{name:7s} big_tiles={big}: {t:6.3f} ms  {2 * S * R * C * N / t / 1e9:6.1f} TF/s", flush=True)
L.csn_dev_set(0, 1)
