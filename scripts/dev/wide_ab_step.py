#!/usr/bin/env python3
"""development aid: the whole training step with the experimental 16-wave GEMM kernel on and off, interleaved.
    python scripts/dev/wide_ab_step.py --config 3 --math bf16x3"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import csn_amd
from csn_amd.csa_models import get_model

CONFIGS = {2: dict(B=4, K=2, N=10000, C=256, nb=20), 3: dict(B=32, K=3, N=10000, C=256, nb=20), 5: dict(B=8, K=4, N=50000, C=96, nb=100)}
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--math", default="bf16x3")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--forms", default="0,1,3,7", help="sets of product forms on the 16-wave kernel (bits: 1 plain, 2 tile-plane B, 4 weight gradients)")
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
    torch.manual_seed(7)
    loss = torch.nn.functional.cross_entropy(model(x, "train", nbf).squeeze(-1), lab, ignore_index=0)
    loss.backward()
    return loss.item()

forms = [int(v) for v in a.forms.split(",")]
L.csn_dev_set(1, 1 if a.math == "bf16x3" else 2)
res, losses, grads = {f: [] for f in forms}, {}, {}
for r in range(a.rounds):
    for f in forms:
        L.csn_dev_set(2, f)
        for _ in range(2):
            losses[f] = step()
        grads[f] = [p.grad.clone() for p in model.parameters() if p.grad is not None]
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        for i in range(5):
            ev[i].record(); step()
        ev[5].record(); torch.cuda.synchronize()
        res[f].append(float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(5)])))
L.csn_dev_set(2, 1)
L.csn_dev_set(1, 1)
for f in forms:
    same = all(torch.equal(x, y) for x, y in zip(grads[f], grads[forms[0]]))
    print(f"config {a.config} {a.math:7s} 16-wave forms {f}: median {np.median(res[f]):7.3f} ms/step  ({' '.join(f'{v:.2f}' for v in res[f])})  loss {losses[f]:.6f}  gradients equal to forms {forms[0]}: {same}", flush=True)
