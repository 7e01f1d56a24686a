#!/usr/bin/env python3
"""Throughput of the MinkowskiNet-variant attention layer (csn_amd/minkowski_attention.py):
one MHA(query shape, key shape, key shape) forward + backward with gradients to every input, the way hrnet.py:378-410 runs
it per shape pair (n_head = 4, d_model = 256: MinkowskiNet/lib/config.py:48-49).  Development aid, not the headline bench."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csn_amd import _lib
from csn_amd.minkowski_attention import MultiHeadAttention


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lq", type=int, default=4001)
    ap.add_argument("--lk", type=int, default=3503)
    ap.add_argument("--pairs", type=int, default=8, help="shape pairs per call (batch dimension)")
    ap.add_argument("--mode", type=int, default=1)
    a = ap.parse_args()
    _lib.check(_lib.lib().csn_set_math_mode(a.mode))
    H, C = 4, 256
    m = MultiHeadAttention(H, C, C // H, C // H, return_attention=False).cuda().train()
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn((a.pairs, a.lq, C), device="cuda", generator=g, requires_grad=True)
    k = torch.randn((a.pairs, a.lk, C), device="cuda", generator=g, requires_grad=True)

    def step():
        for p in m.parameters():
            p.grad = None
        q.grad = k.grad = None
        out, _ = m(q, k, k)
        out.square().mean().backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    D = C
    fwd = a.pairs * (2 * C * D * (a.lq + 2 * a.lk) + 4 * a.lq * a.lk * D + 2 * a.lq * D * C)
    print(f"mode {'bf16x3' if a.mode else 'fp32'}: {a.pairs} pairs, {a.lq} x {a.lk} points, H={H}, d_model={C}: {ms:7.3f} ms per fwd+bwd "
          f"(train mode, gradients to q, k, v and weights) = {a.pairs * a.lq / ms / 1e3:6.2f} M query points/s, "
          f"{3 * fwd / ms / 1e9:6.1f} TFLOP/s algorithmic (3 x forward matmul FLOPs) = {3 * fwd / ms / 1e9 / (2500.0 if a.mode else 157.3):.4f} of the "
          f"{'2.5 PFLOP/s bf16 matrix peak (the pipe sees 3 x that: three products per FLOP)' if a.mode else '157.3 TFLOP/s fp32 matrix peak'}; "
          f"attention products alone (4 Lq Lk d per pair and pass, x 3 passes): {3 * a.pairs * 4.0 * a.lq * a.lk * D / ms / 1e9:6.1f} TFLOP/s")
    _lib.lib().csn_set_math_mode(0)


if __name__ == "__main__":
    main()
