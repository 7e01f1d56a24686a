import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
from csn_amd import _lib, functional as CF
from microbench import timeit
L = _lib.lib(); L.csn_set_math_mode(1)
E, C, D, NP = 128, 256, 256, 10000
att = torch.randn((E, D, NP), device="cuda"); w = torch.randn((C, D), device="cuda") / 16; x = torch.randn((32, C, NP), device="cuda")
ridx = (torch.arange(E, device="cuda", dtype=torch.int32) % 32)
xhat = torch.empty((E, C, NP), device="cuda"); rstd = torch.empty((E, NP), device="cuda")
def f(p):
    _lib.check(L.csn_outproj_ln_fwd_f32(CF._ptr(att), D * NP, CF._ptr(w), CF._ptr(x), C * NP, CF._ptr(ridx), CF._ptr(xhat), C * NP, CF._ptr(rstd), E, C, D, NP, NP, 1e-6, p, 1234, None, None, 0, CF._stream()))
for p in (0.0, 0.1):
    t = timeit(lambda: f(p), n=9)
    print(f"outproj+LN E={E} dropout={p}: {t:6.3f} ms {2*E*C*D*NP/t/1e9:6.1f} TF/s")
