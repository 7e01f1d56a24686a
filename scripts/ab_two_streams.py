#!/usr/bin/env python3
"""Round-6 host-side experiment (a): the training step as TWO HALF-BATCHES ON TWO HIP STREAMS against the single-call step.

The shapes of a step are independent (MID-FC/csa_models.py:209-242 has no term across query shapes) — only the weight
gradients and the loss mean couple them.  Variant `two`: shapes 0..B/2-1 and B/2..B-1 run forward + masked cross-entropy +
backward on a stream each, through two modules that share the weight VALUES but own their gradient slabs (summed once at the
end, weighted by the halves' valid-label counts so that the result is the whole batch's mean loss gradient).  The hypothesis
(round-5 review): the latency-bound attention launches of one half (2.9 / 4.2 TB/s) run beside the streaming launches of the
other (5.5 TB/s).  Variant `one`: the production step.  Interleaved in one process; the gradients are compared (not bit for
bit: the halves' weight-gradient sums associate differently).

    python scripts/ab_two_streams.py [--config 3] [--math bf16x3] [--rounds 3]
Also prints each half's own span (HIP events at the head and tail of its stream) when the two run side by side, against the
span of a half running alone."""
import argparse
import copy
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
ap.add_argument("--steps", type=int, default=7)
a = ap.parse_args()
c = CONFIGS[a.config]
B, K, N, C, nb = c["B"], c["K"], c["N"], c["C"], c["nb"]
L = csn_amd.lib()
csn_amd._lib.check(L.csn_set_math_mode({"fp32": 0, "bf16x3": 1, "bf16": 2, "fp16": 3}[a.math]))
torch.manual_seed(0)
model = get_model("csa", 39, 1, K, d_model=C, d_k=C, d_v=C, block=500, n_blocks=nb).cuda().train()
model.trust_neighbor_slot0 = True
halves = [copy.deepcopy(model) for _ in range(2)]           # same weight values, own gradient slabs
rng = np.random.default_rng(1)
nbf = torch.from_numpy(rng.standard_normal((B, K + 1, C, N, 1)).astype(np.float32)).cuda()
x = nbf[:, 0].contiguous()
lab = torch.from_numpy(np.where(rng.random(size=(B, N)) < 0.1, 0, rng.integers(0, 39, size=(B, N)))).cuda()
h = B // 2
parts = [(x[:h].contiguous(), nbf[:h].contiguous(), lab[:h].contiguous()), (x[h:].contiguous(), nbf[h:].contiguous(), lab[h:].contiguous())]
counts = [float((p[2] > 0).sum()) for p in parts]
weights = [n / sum(counts) for n in counts]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
names = [n for n, p in model.named_parameters() if not n.startswith("fc_1")]


def step_one():
    for p in model.parameters():
        p.grad = None
    torch.manual_seed(7)
    loss = masked_cross_entropy(model(x, "train", nbf), lab, 0)[0]
    loss.backward()
    return loss, [dict(model.named_parameters())[n].grad for n in names]


spans = []                                              # [(start, end) per stream] of the steps that asked for them


def step_two(record=False):
    main = torch.cuda.current_stream()
    torch.manual_seed(7)
    losses, evs = [], []
    for m, s, (xa, na, la), w in zip(halves, streams, parts, weights):
        for p in m.parameters():
            p.grad = None
        s.wait_stream(main)
        with torch.cuda.stream(s):                     # forward, loss AND backward under the half's own ambient stream
            if record:
                e0 = torch.cuda.Event(enable_timing=True); e0.record()
            loss = masked_cross_entropy(m(xa, "train", na), la, 0)[0] * w
            loss.backward()
            losses.append(loss.detach())
            if record:
                e1 = torch.cuda.Event(enable_timing=True); e1.record()
                evs.append((e0, e1))
    if record:
        spans.append(evs)
    for s in streams:
        main.wait_stream(s)
    g = [dict(halves[0].named_parameters())[n].grad + dict(halves[1].named_parameters())[n].grad for n in names]
    return losses[0] + losses[1], g


def timed(fn):
    for _ in range(2):
        loss, g = fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    for i in range(a.steps):
        ev[i].record()
        fn()
    ev[a.steps].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps)])), float(loss), [t.clone() for t in g]


res = {"one": [], "two": []}
out = {}
for r in range(a.rounds):
    for name, fn in (("one", step_one), ("two", step_two)):
        ms, loss, g = timed(fn)
        res[name].append(ms)
        out[name] = (loss, g)
# the halves alone, one after the other on one stream: what two streams would have to beat
seq = []
for r in range(a.rounds):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(a.steps):
        e0.record()
        for m, (xa, na, la), w in zip(halves, parts, weights):
            for p in m.parameters():
                p.grad = None
            (masked_cross_entropy(m(xa, "train", na), la, 0)[0] * w).backward()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    seq.append(float(np.median(ts)))
for _ in range(3):
    step_two(record=True)
torch.cuda.synchronize()
side = [float(np.median([sp[i][0].elapsed_time(sp[i][1]) for sp in spans])) for i in range(2)]
print(f"config {a.config} {a.math}: span of each half side by side {side[0]:.3f} / {side[1]:.3f} ms; a half alone {np.median(seq) / 2:.3f} ms")
g1, g2 = out["one"][1], out["two"][1]
rel = max(((u - v).abs().max() / v.abs().max().clamp_min(1e-30)).item() for u, v in zip(g2, g1))
print(f"config {a.config} {a.math}: one stream, one call   median {np.median(res['one']):7.3f} ms/step  ({' '.join(f'{v:.2f}' for v in res['one'])})  loss {out['one'][0]:.6f}")
print(f"config {a.config} {a.math}: two halves, one stream median {np.median(seq):7.3f} ms/step  ({' '.join(f'{v:.2f}' for v in seq)})")
print(f"config {a.config} {a.math}: two halves, two streams median {np.median(res['two']):7.3f} ms/step  ({' '.join(f'{v:.2f}' for v in res['two'])})  loss {out['two'][0]:.6f}  "
      f"gradients vs one call: max rel diff {rel:.1e} (different masks per half: the seeds are drawn in another order — compare the loss scale, not bits)")
