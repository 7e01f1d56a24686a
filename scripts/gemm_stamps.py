#!/usr/bin/env python3
"""development aid: prologue / main loop / epilogue cycles of the bf16x3 GEMM (build with -DCSN_STAMPS into build/gs.so)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CSN_LIB_PATH", "build/gs.so")
from csn_amd import _lib, functional as CF
L = _lib.lib()
L.csn_set_math_mode(1)
L.csn_gemm_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
S, C, N, R = 32, 256, 10000, 768
x = torch.randn((S, C, N), device="cuda"); w = torch.randn((R, C), device="cuda") / 16
dout = torch.randn((S, R, N), device="cuda")

def report(name, nwg, nslab):
    torch.cuda.synchronize()
    buf = np.zeros(65536 * 8, dtype=np.uint64)
    L.csn_gemm_debug_read(buf.ctypes.data, buf.nbytes)
    full = buf.reshape(65536, 8)[:min(nwg, 65536)].astype(np.int64)
    st, inner = full[:, :4], full[:, 4:]
    print(f"   per slab: reads+mfma={inner[:,0].mean()/nslab:6.0f}  stage(split+lds write+issue loads)={inner[:,1].mean()/nslab:6.0f}  barrier={inner[:,2].mean()/nslab:6.0f}")
    d = np.diff(st, axis=1)
    t0 = st[:, 0].min()
    span = st[:, 3].max() - t0
    print(f"   span={span} cycles, sum of WG lifetimes / span / 256 CUs = {(st[:,3]-st[:,0]).sum() / span / 256:5.2f} concurrent WGs per CU")
    print(f"{name}: work-groups {nwg}  prologue={d[:,0].mean():7.0f}  loop={d[:,1].mean():7.0f} ({d[:,1].mean()/nslab:6.0f}/slab x {nslab})  "
          f"epilogue={d[:,2].mean():7.0f}  wg total={(st[:,3]-st[:,0]).mean():7.0f}  kernel span={(st[:,3].max()-t0)}  "
          f"concurrency={(st[:,3]-st[:,0]).sum()/(st[:,3].max()-t0)/256:5.2f} WG/CU")

for _ in range(2): CF.project(x, w)
report("project KN 768x10000x256 x32", 32 * 6 * 79, 8)

