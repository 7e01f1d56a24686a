#!/usr/bin/env python3
"""development aid: per-phase cycle stamps of the key-stationary dK / dV kernel at config-5 geometry (d = 96, one plane).
Library: scripts/dev/build_variant_one.sh dst attn_dkv.hip -DCSN_DKV_STAMPS -DCSN_DKV_PIPE=0   (CSN_LIB_PATH picks it up)."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("CSN_LIB_PATH", "csn_amd/libcsn_dst.so")
from csn_amd import _lib, functional as CF
from bench_attn import to_tiles
L = _lib.lib()
RAW = L._handle
H, d, T, nb = 1, 96, 500, 100
D, NP, Tp, S, E = 96, 50000, 512, 40, 80
L.csn_set_math_mode(2)
qkv = torch.randn((S, 3 * D, NP), device="cuda"); qkv[:, :D] *= 0.25
datt = torch.randn((E, D, NP), device="cuda")
qs = torch.arange(E, device="cuda", dtype=torch.int32) % S
ks = (torch.arange(E, device="cuda", dtype=torch.int32) * 7 + 3) % S
lse = torch.randn((E, H, NP), device="cuda") + 6.0
delta = torch.randn((E, H, NP), device="cuda") * 0.01
dqkv = torch.zeros((E, 3 * D, NP), device="cuda")
kvt = to_tiles(qkv[:, D:], T, nb, 1)
k_ptr, kvp = kvt.data_ptr(), nb * 512
v_ptr, kv_stride = k_ptr + 2 * D * kvp, 2 * D * kvp
gb = dqkv.data_ptr()
drop = float(os.environ.get("CSN_STAMP_DROP", "0.1"))
A16 = int(os.environ.get("CSN_STAMP_ACT16", "2"))     # 0: fp32 maps; 2: Qs fp16 / dO bf16 maps (the fp16 configuration's backward)
if A16:
    q16 = qkv[:, :D].to(torch.float16 if A16 == 2 else torch.bfloat16).contiguous()
    d16 = datt.to(torch.bfloat16).contiguous()
    L.csn_set_thread_act16(A16)


def dkv():
    _lib.check(L.csn_block_attn_bwd_dkv_flash_f32(CF._ptr(d16 if A16 else datt), D * NP, q16.data_ptr() if A16 else qkv.data_ptr(), D * NP if A16 else 3 * D * NP, CF._ptr(qs), k_ptr, v_ptr, kv_stride,
                                                  CF._ptr(ks), kvp, 0, NP, CF._ptr(lse), CF._ptr(delta), gb + 4 * D * NP, gb + 8 * D * NP,
                                                  3 * D * NP, None, None, 0, None, E, H, d, T, nb, Tp, drop, 1234, None, 0, CF._stream()), "dkv flash")


t_end, n = time.time() + 2.0, 0
while n < 2 or time.time() < t_end:                  # sustained load first: the clock settles
    n += 1
    dkv()
    if n % 8 == 0:
        torch.cuda.synchronize()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); dkv(); e1.record(); torch.cuda.synchronize()
print(f"launch {e0.elapsed_time(e1):.3f} ms (instrumented build)")
cnt = 1024 * 8 * 4 * 8
buf = np.zeros(cnt, dtype=np.uint64)
RAW.csn_dkv_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
rc = RAW.csn_dkv_debug_read(buf.ctypes.data, cnt * 8)
st = buf.reshape(1024, 8, 4, 8).astype(np.int64)
ok = (st[..., 0] > 0).all(axis=(1, 2))
dd = np.diff(st, axis=-1)[ok]
names = ["phase1", "commitK+fetch", "barrier1", "pointwise", "phase2", "commitC", "barrier2"]
print("rc", rc, "work-groups with stamps", int(ok.sum()))
for w in (0, 1, 4, 5):
    print(f"wave {w}: " + "  ".join(f"{nm}={dd[:, w, :, i].mean():6.0f}" for i, nm in enumerate(names)), f" tile={dd[:, w].sum(axis=-1).mean():6.0f}")
print("early half: " + "  ".join(f"{nm}={dd[:, :4, :, i].mean():6.0f}" for i, nm in enumerate(names)), f" tile={dd[:, :4].sum(axis=-1).mean():6.0f}")
print("late half:  " + "  ".join(f"{nm}={dd[:, 4:, :, i].mean():6.0f}" for i, nm in enumerate(names)), f" tile={dd[:, 4:].sum(axis=-1).mean():6.0f}")
# tile-to-tile period from stamp 0 of consecutive tiles
per = np.diff(st[ok][..., 0], axis=-1)
print(f"tile period (stamp 0 to stamp 0): {per.mean():.0f} cycles")
