#!/usr/bin/env python3
"""The whole training step (forward + masked cross-entropy + backward, train mode) under several settings of the library's
development switches (csn_dev_set, include/csn_hip.h), interleaved in one process: median ms per step per variant, the loss, and
whether the gradients equal those of the first variant bit for bit.
    python scripts/ab_step.py --config 3 --math bf16x3 --variants "tiled:3=0;streaming:3=1"      (3 = CSN_DEV_WX)
A setting `name=value` with a non-numeric name flips a switch of csn_amd.tuning instead ("rows:tile_major_scores=0;tiles:")."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import csn_amd  # noqa: E402
from csn_amd.csa_models import get_model  # noqa: E402
from csn_amd.functional import masked_cross_entropy  # noqa: E402

CONFIGS = {2: dict(B=4, K=2, N=10000, C=256, nb=20), 3: dict(B=32, K=3, N=10000, C=256, nb=20), 5: dict(B=8, K=4, N=50000, C=96, nb=100)}
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--math", default="bf16x3")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--variants", default="tiled:3=0;streaming:3=1", help="name:key=value,key=value;name:...")
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
    loss = masked_cross_entropy(model(x, "train", nbf), lab, 0)[0]
    loss.backward()
    return loss.item()


variants = []
for spec in a.variants.split(";"):
    name, _, kv = spec.partition(":")
    items = [item.split("=") for item in kv.split(",") if item]
    variants.append((name, [(int(k), int(v)) for k, v in items if k.isdigit()], {k: bool(int(v)) for k, v in items if not k.isdigit()}))
defaults = {k: L.csn_dev_get(k) for _, kvs, _ in variants for k, _ in kvs}
from csn_amd import tuning  # noqa: E402
res, losses, grads = {n: [] for n, _, _ in variants}, {}, {}
for r in range(a.rounds):
  for name, kvs, tun in variants:
    with tuning.override(**tun):
        for k, v in defaults.items():
            L.csn_dev_set(k, v)
        for k, v in kvs:
            L.csn_dev_set(k, v)
        for _ in range(2):
            losses[name] = step()
        grads[name] = [p.grad.clone() for p in model.parameters() if p.grad is not None]
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        for i in range(5):
            ev[i].record()
            step()
        ev[5].record()
        torch.cuda.synchronize()
        res[name].append(float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(5)])))
for k, v in defaults.items():
    L.csn_dev_set(k, v)
first = variants[0][0]
for name, _, _ in variants:
    same = all(torch.equal(g, h) for g, h in zip(grads[name], grads[first]))
    rel = [((g - h).abs().max() / h.abs().max().clamp_min(1e-30)).item() for g, h in zip(grads[name], grads[first])]
    worst = max(rel)
    pname = [n for n, q in model.named_parameters() if q.grad is not None][int(np.argmax(rel))]
    print(f"config {a.config} {a.math:7s} {name:>16s}: median {np.median(res[name]):7.3f} ms/step  ({' '.join(f'{v:.2f}' for v in res[name])})  "
          f"loss {losses[name]:.6f}  gradients vs {first}: {'bit-equal' if same else f'max rel diff {worst:.1e} ({pname}, median over parameters {np.median(rel):.1e})'}", flush=True)
