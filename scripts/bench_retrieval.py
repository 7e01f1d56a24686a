#!/usr/bin/env python3
"""Throughput of the retrieval measure behind the kNN shape graph (csn_retrieval_measure_f32; csa_models.py:244-267):
r[i, j] = mean_n max_m cos(f_i[n], f_j[m]) for S x S shape pairs of N points, always in exact fp32."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csn_amd import functional as CF

S, N, C = 16, 10000, 256
f = torch.randn((S, N, C), device="cuda")
CF.retrieval_measure(f, f); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    r = CF.retrieval_measure(f, f)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 3 * 1e3
fl = 2.0 * S * S * N * N * C
tf = fl / ms / 1e9
print(f"retrieval measure {S} x {S} shapes of {N} points: {ms:8.2f} ms = {S * S / ms * 1e3:7.1f} shape pairs/s, {tf:6.1f} TFLOP/s algorithmic "
      f"(2 N^2 C = {2.0 * N * N * C / 1e9:.1f} GFLOP per pair) = {tf / 157.3:.3f} of the 157.3 TFLOP/s fp32 matrix peak, diag {r.diag().mean().item():.6f}")
