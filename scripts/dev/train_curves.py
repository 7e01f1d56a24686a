#!/usr/bin/env python3
"""development aid: the same short training run (same initial weights, data, dropout seeds, Adam) in each math mode — do the
16-bit modes with the 16-bit exchange follow the bf16x3 trajectory?   python scripts/dev/train_curves.py [--steps 40]"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import csn_amd
import csn_amd.functional
from csn_amd.csa_models import get_model

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=40)
a = ap.parse_args()
B, K, N, C, n_cls = 4, 2, 2000, 256, 12
rng = np.random.default_rng(5)
# a learnable task: the label of a point is a function of its own features (argmax over a random projection)
proj = rng.standard_normal((n_cls, C)).astype(np.float32)
feats = rng.standard_normal((B, K + 1, C, N, 1)).astype(np.float32)
lab = 1 + np.argmax(np.einsum("kc,bcn->bkn", proj[1:], feats[:, 0, :, :, 0]), axis=1)
nbf, lab = torch.from_numpy(feats).cuda(), torch.from_numpy(lab).cuda()
curves = {}
for math in ("fp32", "bf16x3", "bf16", "fp16"):
    torch.manual_seed(0)
    model = get_model("csa", n_cls, 1, K, block=500, n_blocks=4, math=math).cuda().train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for it in range(a.steps):
        torch.manual_seed(100 + it)
        opt.zero_grad()
        loss = csn_amd.functional.masked_cross_entropy(model(nbf[:, 0], "train", nbf), lab, 0)[0]
        loss.backward()
        opt.step()
        losses.append(loss.item())
    curves[math] = losses
    assert all(np.isfinite(losses))
print(f"# CSA layer + logit head, B={B} K={K} N={N} C={C}, {n_cls} classes, Adam 1e-3, dropout live, same seeds in every mode; loss per step")
print("step  " + "  ".join(f"{m:>9s}" for m in curves))
for it in list(range(0, a.steps, max(1, a.steps // 10))) + [a.steps - 1]:
    print(f"{it:4d}  " + "  ".join(f"{curves[m][it]:9.5f}" for m in curves))
ref = np.array(curves["fp32"])
for m in curves:
    print(f"# {m:7s}: final loss {curves[m][-1]:.5f}, max |loss - fp32 loss| over the run {np.abs(np.array(curves[m]) - ref).max():.2e}")
