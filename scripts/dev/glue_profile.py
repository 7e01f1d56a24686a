#!/usr/bin/env python3
"""development aid: which torch ops of one config-3 training step are not libcsn kernels (torch.profiler, self device time)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import csn_amd
from csn_amd.csa_models import get_model
from torch.profiler import profile, ProfilerActivity

B, K, N, C, n_cls = 32, 3, 10000, 256, 39
torch.manual_seed(0)
model = get_model("csa", n_cls, 1, K).cuda().train()
model.trust_neighbor_slot0 = True
rng = np.random.default_rng(1)
nb = torch.from_numpy(rng.standard_normal((B, K + 1, C, N, 1)).astype(np.float32)).cuda()
x = nb[:, 0].contiguous()
lab = torch.from_numpy(rng.integers(0, n_cls, size=(B, N))).cuda()

def step():
    for p in model.parameters():
        p.grad = None
    logits = model(x, "train", nb)
    loss = torch.nn.functional.cross_entropy(logits.squeeze(-1), lab, ignore_index=0)
    loss.backward()

for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = e.self_cuda_time_total
    if t > 0:
        rows.append((t, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"ops with device time: {len(rows)}, total {tot / 1e3:.3f} ms")
for t, c, k, sh in rows[:70]:
    print(f"{t / 1e3:8.3f} ms  x{c:<3d} {k:38s} {sh}")
