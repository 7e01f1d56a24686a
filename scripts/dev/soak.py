#!/usr/bin/env python3
"""development aid: many training steps of the config-3 workload in one process — allocator high-water mark and step time must
stay flat (no growth of saved state across steps)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import csn_amd
from csn_amd.csa_models import get_model
from oracle import csa_oracle as orc            # labels / loss only (development script)

csn_amd.build()
B, K, N, C, n_cls = 32, 3, 10000, 256, 39
rng = np.random.default_rng(0)
model = get_model("csa", n_cls, 1, K).cuda().train()
model.trust_neighbor_slot0 = True
x = torch.randn(B, C, N, 1, device="cuda")
nb = torch.randn(B, K + 1, C, N, 1, device="cuda"); nb[:, 0] = x
lab = orc.synth_labels(rng, B, N, n_cls).cuda()
opt = torch.optim.Adam([p for n, p in model.named_parameters() if not n.startswith("fc_1")], lr=1e-4, betas=(0.5, 0.999))
marks = []
for rnd in range(6):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(50):
        opt.zero_grad(set_to_none=True)
        loss = orc.masked_ce_loss(model(x, "train", nb), lab)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    marks.append(((time.time() - t0) / 50 * 1e3, torch.cuda.memory_allocated() / 2**30, torch.cuda.max_memory_allocated() / 2**30, loss.item()))
    print(f"round {rnd}: {marks[-1][0]:.2f} ms/step, allocated {marks[-1][1]:.2f} GiB, peak {marks[-1][2]:.2f} GiB, loss {marks[-1][3]:.4f}", flush=True)
assert abs(marks[-1][1] - marks[1][1]) < 0.01 and abs(marks[-1][2] - marks[1][2]) < 0.01, "allocator state grows across steps"
assert np.isfinite(marks[-1][3]) and marks[-1][3] < marks[0][3], "the loss does not go down"
print("ok")
