#!/usr/bin/env python3
"""development aid: the whole training step in each attention-backward data flow, interleaved in one process.
    python scripts/dev/flow_ab_step.py --config 3 --math bf16"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import csn_amd
import csn_amd.functional
from csn_amd import tuning
from csn_amd.csa_models import get_model

CONFIGS = {2: dict(B=4, K=2, N=10000, C=256, nb=20), 3: dict(B=32, K=3, N=10000, C=256, nb=20), 5: dict(B=8, K=4, N=50000, C=96, nb=100)}
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--math", default="bf16")
ap.add_argument("--rounds", type=int, default=4)
a = ap.parse_args()
c = CONFIGS[a.config]
B, K, N, C, nb = c["B"], c["K"], c["N"], c["C"], c["nb"]
mode = {"fp32": 0, "bf16x3": 1, "bf16": 2, "fp16": 3}[a.math]
csn_amd._lib.check(csn_amd.lib().csn_set_math_mode(mode))
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
    loss = csn_amd.functional.masked_cross_entropy(model(x, "train", nbf), lab, 0)[0]
    loss.backward()

bits = csn_amd.lib().csn_attn_bwd_grouping(C, 500) if mode != 3 else 0
if mode == 3:
    csn_amd.lib().csn_set_math_mode(2); bits = csn_amd.lib().csn_attn_bwd_grouping(C, 500); csn_amd.lib().csn_set_math_mode(3)
flows = [("keep", tuning.KEEP_SCORES)] + ([("recompute_dq", tuning.RECOMPUTE_DQ)] if bits & 4 else []) + ([("flash", tuning.FLASH)] if bits & 8 else [])
res = {n: [] for n, _ in flows}
for r in range(a.rounds):
    for name, f in flows:
        with tuning.override(score_flow={1: f, 2: f}):
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
            for i in range(5):
                ev[i].record(); step()
            ev[5].record(); torch.cuda.synchronize()
            res[name].append(float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(5)])))
for n in res:
    print(f"config {a.config} {a.math:7s} flow {n:13s}: median {np.median(res[n]):7.3f} ms/step  (rounds: {' '.join(f'{v:.2f}' for v in res[n])})", flush=True)
