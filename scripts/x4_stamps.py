#!/usr/bin/env python3
"""Where a key tile of the four-wave attention forward (attn_fwd_x4.hip) spends its cycles: builds the library with
-DCSN_X4_STAMPS into a scratch directory, runs the forward at config-3 geometry and prints the mean cycles of every part of
tiles 4..11 (s_memtime at the part boundaries; shares, not lengths: the stamps' fences forbid overlaps the product build has)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from csn_amd import _lib  # noqa: E402

out = "/tmp/csn_x4_stamps"
os.makedirs(out, exist_ok=True)
so = os.path.join(out, "libcsn_hip.so")
srcs = [os.path.join(ROOT, "csn_amd", "csrc", f) for f in _lib.SOURCES]
subprocess.run(["/opt/rocm/bin/hipcc", *_lib.BUILD_FLAGS, "-DCSN_X4_STAMPS", *srcs, "-o", so], check=True)
os.environ["CSN_LIB_PATH"] = so
L = _lib.lib()
_lib.check(L.csn_set_math_mode(1))
sys.argv = [sys.argv[0], "--tiles", "--only", "fwd", "--evals", "64", "--drop", "0.1"]
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import bench_attn  # noqa: E402
bench_attn.main()
h = ctypes.CDLL(so)
buf = np.zeros(256 * 4 * 8 * 8, dtype=np.uint64)
h.csn_x4_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
h.csn_x4_debug_read(buf.ctypes.data, buf.nbytes)
st = buf.reshape(256, 4, 8, 8).astype(np.int64)
ok = st[..., 0] > 0
d = np.diff(st[..., :7], axis=-1)[ok]
names = ["issue 16 DMA pieces", "mask / max / rescale check", "S(t+1) matrix || pointwise(t)", "O += V P matrix", "wait for the DMA", "barrier"]
tot = d.sum(axis=1).mean()
print(f"tiles stamped: {ok.sum()}, mean cycles per tile {tot:.0f} (48 + 48 matrix instructions = 3072 pipe cycles)")
for i, n in enumerate(names):
    print(f"  {n:34s} {d[:, i].mean():8.0f}  ({100 * d[:, i].mean() / tot:4.1f} %)   min {d[:, i].min():6d}  max {d[:, i].max():6d}")
