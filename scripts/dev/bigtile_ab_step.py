#!/usr/bin/env python3
"""development aid: the whole training step with the 256 x 256 GEMM tiles on and off (csn_debug_set_big_tiles), interleaved.
    python scripts/dev/bigtile_ab_step.py --config 5 --math fp16"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import csn_amd
from csn_amd.csa_models import get_model

CONFIGS = {2: dict(B=4, K=2, N=10000, C=256, nb=20), 3: dict(B=32, K=3, N=10000, C=256, nb=20), 5: dict(B=8, K=4, N=50000, C=96, nb=100)}
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=5)
ap.add_argument("--math", default="fp16")
ap.add_argument("--rounds", type=int, default=4)
a = ap.parse_args()
c = CONFIGS[a.config]
B, K, N, C, nb = c["B"], c["K"], c["N"], c["C"], c["nb"]
L = csn_amd.lib()
csn_amd._lib.check(L.csn_set_math_mode({"fp32": 0, "bf16x3": 1, "bf16": 2, "fp16": 3}[a.math]))
torch.manual_seed(0)
model = get_model("csa", 39, 1, K, d_model=C, d_k=C, d_v=C, block=500, n_blocks=nb).cuda().train()
model.trust_neighbor_slot0 = True
rng = np.random.default_rng(1)
nbf = torch.from_numpy(rng.standard_normal((B, K + 1, C, N, 1)).astype(np.float32)).cuda()
x = nbf[:, 0].contiguous()
lab = torch.from_numpy(rng.integers(0, 39, size=(B, N))).cuda()

def step():
    for p in model.parameters():
        p.grad = None
    torch.nn.functional.cross_entropy(model(x, "train", nbf).squeeze(-1), lab, ignore_index=0).backward()

res = {0: [], 1: []}
for r in range(a.rounds):
    for on in (1, 0):
        L.csn_dev_set(0, on)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        for i in range(5):
            ev[i].record(); step()
        ev[5].record(); torch.cuda.synchronize()
        res[on].append(float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(5)])))
L.csn_dev_set(0, 1)
for on in (1, 0):
    print(f"config {a.config} {a.math:7s} 256x256 tiles {'on ' if on else 'off'}: median {np.median(res[on]):7.3f} ms/step  ({' '.join(f'{v:.2f}' for v in res[on])})", flush=True)
