#!/usr/bin/env python3
"""development aid: phase stamps of the 256 x 256 bf16x3 GEMM (library built with -DCSN_STAMPS into build/gs.so):
out-projection + LayerNorm forward, and the plain projection."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CSN_LIB_PATH", "build/gs.so")
from csn_amd import _lib, functional as CF
L = _lib.lib()
L.csn_set_math_mode(1)
L.csn_gemm_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
E, C, D, NP = 128, 256, 256, 10000
att = torch.randn((E, D, NP), device="cuda"); w = torch.randn((C, D), device="cuda") / 16; x = torch.randn((32, C, NP), device="cuda")
ridx = (torch.arange(E, device="cuda", dtype=torch.int32) % 32)
xhat = torch.empty((E, C, NP), device="cuda"); rstd = torch.empty((E, NP), device="cuda")


def report(name, labels):
    torch.cuda.synchronize()
    buf = np.zeros(65536 * 8, dtype=np.uint64)
    L.csn_gemm_debug_read(buf.ctypes.data, buf.nbytes)
    st = buf.reshape(65536, 8)[2048:5120].astype(np.int64)          # work-groups well inside the launch
    d = np.diff(st[:, :len(labels) + 1], axis=1)
    print(name, " ".join(f"{l}={d[:, i].mean():8.0f}" for i, l in enumerate(labels)), f" total={(st[:, len(labels)] - st[:, 0]).mean():8.0f}")


for _ in range(3):
    _lib.check(L.csn_outproj_ln_fwd_f32(CF._ptr(att), D * NP, CF._ptr(w), CF._ptr(x), C * NP, CF._ptr(ridx), CF._ptr(xhat), C * NP,
                                        CF._ptr(rstd), E, C, D, NP, NP, 1e-6, 0.1, 1234, None, None, 0, CF._stream()))
report("out-projection + LN:", ["setup", "tile loop", "residual + mean", "variance", "normalise + store"])
# the same launch with every evaluation reading the SAME context map and residual (cache-resident operands): what the phases
# cost when HBM is out of the picture
ridx0 = torch.zeros_like(ridx)
for _ in range(3):
    _lib.check(L.csn_outproj_ln_fwd_f32(CF._ptr(att), 0, CF._ptr(w), CF._ptr(x), C * NP, CF._ptr(ridx0), CF._ptr(xhat), C * NP,
                                        CF._ptr(rstd), E, C, D, NP, NP, 1e-6, 0.1, 1234, None, None, 0, CF._stream()))
report("  ... cache-resident inputs:", ["setup", "tile loop", "residual + mean", "variance", "normalise + store"])
xs = torch.randn((128, C, NP), device="cuda"); wq = torch.randn((256, C), device="cuda") / 16
for _ in range(3):
    CF.project(xs, wq)
report("projection 256 rows:", ["setup", "tile loop"] ) if False else None
torch.cuda.synchronize()
buf = np.zeros(65536 * 8, dtype=np.uint64)
L.csn_gemm_debug_read(buf.ctypes.data, buf.nbytes)
st = buf.reshape(65536, 8)[2048:5120].astype(np.int64)
print("projection 256 rows: setup=%8.0f tile loop=%8.0f epilogue=%8.0f total=%8.0f" % ((st[:, 1] - st[:, 0]).mean(), (st[:, 2] - st[:, 1]).mean(), (st[:, 5] - st[:, 2]).mean(), (st[:, 5] - st[:, 0]).mean()))
