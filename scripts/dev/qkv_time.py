import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from csn_amd import _lib
_lib.build(); L = _lib.lib(); _lib.check(L.csn_set_math_mode(1))
S, C, D, N, T, nb = 128, 256, 256, 10000, 500, 20
ldp = nb * 1024
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn((S, C, N), device="cuda", generator=g); w = torch.randn((768, C), device="cuda", generator=g) / 16
q = torch.empty((S, D, N), device="cuda"); kv = torch.empty((S, 2 * D, ldp), device="cuda", dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
def one(): _lib.check(L.csn_project_qkv_f32(x.data_ptr(), C * N, N, w.data_ptr(), D, C, q.data_ptr(), D * N, N, kv.data_ptr(), 2 * D * ldp, ldp, S, N, 16.0, T, st))
def two():
    _lib.check(L.csn_project_f32(x.data_ptr(), C * N, N, w.data_ptr(), D, C, q.data_ptr(), D * N, N, S, N, D, 16.0, 0, 0, st))
    _lib.check(L.csn_project_f32(x.data_ptr(), C * N, N, w[D:].data_ptr(), 2 * D, C, kv.data_ptr(), 2 * D * ldp, ldp, S, N, 0, 1.0, 2, T, st))
ts = {"one": [], "two": []}
for rep in range(12):
    for name, fn in (("one", one), ("two", two)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if rep >= 2: ts[name].append(e0.elapsed_time(e1))
print("one pass %.3f ms   two calls %.3f ms" % (np.median(ts["one"]), np.median(ts["two"])))
