#!/usr/bin/env python3
"""development aid: per-phase cycle stamps of the instrumented attention build (build/v9.so, -DCSN_STAMPS)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("CSN_LIB_PATH", "csn_amd/libcsn_stf.so")      # scripts/dev/build_variant.sh stf -DCSN_STAMPS=0
from csn_amd import _lib, functional as CF
from bench_attn import to_tiles
L = _lib.lib()
RAW = L._handle          # the stamp readers are not part of the declared interface
H, d, T, nb = 1, 256, 500, 20
D, NP, Tp, S, E = 256, 10000, 512, 16, 64
qkv = torch.randn((S, 3 * D, NP), device="cuda"); qkv[:, :D] *= 0.25
qs = torch.arange(E, device="cuda", dtype=torch.int32) % S
ks = (torch.arange(E, device="cuda", dtype=torch.int32) * 7 + 3) % S
att = torch.empty((E, D, NP), device="cuda"); lse = torch.empty((E, H, NP), device="cuda")
scores = torch.empty((E, H, nb, T, Tp), device="cuda")
base = qkv.data_ptr()
MODE = int(os.environ.get("CSN_STAMP_MODE", "1"))      # 1: bf16x3 (two planes per tile), 2: bf16 (one plane)
BP = 1024 if MODE == 1 else 512                        # tile-plane block pitch
if MODE == 1:
    kvt = to_tiles(qkv[:, D:], T, nb)
else:
    x = torch.zeros((S, 2 * D, nb, 512), device="cuda"); x[..., :T] = qkv[:, D:].view(S, 2 * D, nb, T)
    kvt = x.bfloat16().reshape(S, 2 * D, nb * 512).contiguous()
kp = kvt.data_ptr()
L.csn_set_math_mode(MODE)
import time
t_end = time.time() + float(os.environ.get("CSN_STAMP_SUSTAIN", "2.0"))     # sustained load first: the clock settles
n_launch = 0
while n_launch < 2 or time.time() < t_end:
    n_launch += 1
    if n_launch % 8 == 0:
        torch.cuda.synchronize()
    _lib.check(L.csn_block_attn_fwd_f32(base, kp, kp + 2 * D * nb * BP, 3 * D * NP, 2 * D * nb * BP, CF._ptr(qs), CF._ptr(ks), NP,
                                        CF._ptr(att), D * NP, CF._ptr(scores), CF._ptr(lse), E, H, d, T, nb, Tp, 8.0, 0.1, 1234, 1, nb * BP,
                                        CF._stream()), "fwd")
if os.environ.get("CSN_STAMP_BWD") == "1":       # library built with -DCSN_STAMPS=1: stamps come from the backward (dq) kernel
    datt = torch.randn((E, D, NP), device="cuda")
    dscores = torch.empty_like(scores); delta = torch.empty((E, H, NP), device="cuda"); dq = torch.empty((E, D, NP), device="cuda")
    _lib.check(L.csn_block_attn_bwd_dq_f32(CF._ptr(datt), CF._ptr(att), D * NP, kp, kp + 2 * D * nb * BP, 2 * D * nb * BP,
                                           CF._ptr(ks), NP, CF._ptr(scores), CF._ptr(dscores), CF._ptr(lse), CF._ptr(delta),
                                           CF._ptr(dq), D * NP, None, 0, None, E, H, d, T, nb, Tp, 0.1, 1234, 0, 0, 1, nb * BP, 1, None, 0,
                                           CF._stream()), "dq")
torch.cuda.synchronize()
n = 2048 * 8 * 4 * 8
buf = np.zeros(n, dtype=np.uint64)
RAW.csn_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
rc = RAW.csn_debug_read(buf.ctypes.data, n * 8)
st = buf.reshape(2048, 8, 4, 8).astype(np.int64)
d_ = np.diff(st, axis=-1)                     # [wg][wave][iter][6 segments]
names = ["P1", "commitA+fetchB", "barrierX", "pointwise", "P2", "commitB+fetchA", "barrierY"]
ok = (st[..., 0] > 0).all(axis=(1, 2))
d_ = d_[ok]
print("rc", rc, "work-groups with stamps", ok.sum())
for w in (0, 4, 3, 7):
    print(f"wave {w}: " + "  ".join(f"{n}={d_[:, w, :, i].mean():7.0f}" for i, n in enumerate(names)),
          f" total={d_[:, w].sum(axis=-1).mean():7.0f}")
print("all waves: " + "  ".join(f"{n}={d_[..., i].mean():7.0f}" for i, n in enumerate(names)), f" total={d_.sum(axis=-1).mean():7.0f}")
# skew between wave 0 and wave 4 at the start of P1
print("start skew wave4 - wave0:", (st[ok][:, 4, :, 0] - st[ok][:, 0, :, 0]).mean(), " end-of-P1 skew:", (st[ok][:, 4, :, 1] - st[ok][:, 0, :, 1]).mean())

# whole-kernel stamps (work-groups 4096..6143, i.e. well inside the launch): entry, loop start, loop end, exit
m = 2048 * 8 * 4
wb = np.zeros(m, dtype=np.uint64)
RAW.csn_debug_read_wg.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
RAW.csn_debug_read_wg(wb.ctypes.data, m * 8)
w = wb.reshape(2048, 8, 4).astype(np.int64)
okw = (w[..., 0] > 0).all(axis=1)
w = w[okw]
dw = np.diff(w, axis=-1)
t0 = w[:, :, 0].min(axis=1, keepdims=True)
print("whole kernel, work-groups:", okw.sum(), " (s_memtime ticks = 100 MHz x ? — same unit as above)")
print("  per wave mean: prologue=%.0f loop=%.0f epilogue=%.0f total=%.0f" % (dw[..., 0].mean(), dw[..., 1].mean(), dw[..., 2].mean(), (w[..., 3] - w[..., 0]).mean()))
print("  per work-group (first entry -> last exit): %.0f" % (w[:, :, 3].max(axis=1) - w[:, :, 0].min(axis=1)).mean())
rb = np.zeros(2048 * 8 * 2, dtype=np.uint64)
RAW.csn_debug_read_rt.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
RAW.csn_debug_read_rt(rb.ctypes.data, rb.nbytes)
r = rb.reshape(2048, 8, 2).astype(np.int64)[okw]
clk = (w[..., 3] - w[..., 0]) / np.maximum(r[..., 1] - r[..., 0], 1) * 100.0
print("  in-kernel clock (d s_memtime / d s_memrealtime x 100 MHz): median %.0f MHz  (p10 %.0f, p90 %.0f);  work-group wall %.1f us" %
      (np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90), np.median(r[..., 1] - r[..., 0]) / 100.0))

# prologue stamps (forward, or backward with CSN_STAMP_BWD=1): entry of the item loop body, operand block requested, landed
# (barrier), picked (forward: before the tile fetches), first tile A fetched + committed, B likewise, loop start
pb = np.zeros(2048 * 8 * 8, dtype=np.uint64)
RAW.csn_debug_read_pro.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
RAW.csn_debug_read_pro(pb.ctypes.data, pb.nbytes)
pr = pb.reshape(2048, 8, 8).astype(np.int64)[okw]
dp = np.diff(pr, axis=-1)
names = ["request operand block", "wait + barrier", "pick + split", "(backward: delta round) ", "tile A fetch + commit", "tile B fetch + commit", "request tile 1 + barrier"]
print("prologue, mean cycles per wave: " + "  ".join(f"{n}={dp[..., i].mean():.0f}" for i, n in enumerate(names)), " sum=%.0f" % (pr[..., 7] - pr[..., 0]).mean())
