#!/usr/bin/env python3
"""Which torch (non-csn) kernels run inside one config-3 training step, with the Python line that launched each: the "glue"
that is left between the library's launches."""
import os
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import csn_amd
import csn_amd.functional  # noqa: E402
from csn_amd.csa_models import get_model  # noqa: E402

B, K, N, C, nb = 32, 3, 10000, 256, 20
torch.manual_seed(0)
model = get_model("csa", 39, 1, K).cuda().train()
model.trust_neighbor_slot0 = True
rng = np.random.default_rng(1)
nbf = torch.from_numpy(rng.standard_normal((B, K + 1, C, N, 1)).astype(np.float32)).cuda()
x = nbf[:, 0].contiguous()
lab = torch.from_numpy(rng.integers(0, 39, size=(B, N))).cuda()
use_fused = len(sys.argv) > 1 and sys.argv[1] == "fused"


def step():
    for p in model.parameters():
        p.grad = None
    logits = model(x, "train", nbf)
    if use_fused:
        from csn_amd.training import masked_ce
        loss = masked_ce(logits, lab)
    else:
        loss = csn_amd.functional.masked_cross_entropy(logits, lab, 0)[0]
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA or getattr(e, "device_time_total", 0) <= 0:
        continue
for e in prof.key_averages(group_by_stack_n=6, group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
    if t > 3 and "csn_" not in e.key:
        stack = [s for s in (e.stack or []) if "csn_amd" in s or "bench" in s or "glue_profile" in s]
        rows.append((t, e.count, e.key[:60], str(e.input_shapes)[:80], (stack[0] if stack else "")[-90:]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"non-csn device time in one step: {tot / 1e3:.3f} ms")
for t, c, k, sh, st in rows[:40]:
    print(f"{t / 1e3:7.3f} ms x{c:<3d} {k:60s} {sh:80s} {st}")
