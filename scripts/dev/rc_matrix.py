#!/usr/bin/env python3
"""development aid: dQ of the kept-scores and the recomputing call, with and without P / dS planes, against each other."""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from csn_amd import _lib as L
from tests.test_gpu_flash import tile_planes, _rand, _stream

lib = L.lib()
for (mode, S, E, H, d, T, nb) in [(1, 1, 1, 1, 64, 64, 1), (1, 2, 3, 1, 128, 500, 2)]:
    L.check(lib.csn_set_math_mode(mode))
    rng = np.random.default_rng(100 + d + T)
    D, N, npl = H * d, T * nb, (2 if mode == 1 else 1)
    Tp = (T + 31) // 32 * 32
    q = (_rand(rng, S, D, N) / math.sqrt(math.sqrt(d))).cuda()
    k = _rand(rng, S, D, N) / math.sqrt(math.sqrt(d)); v = _rand(rng, S, D, N)
    dctx = _rand(rng, E, D, N).cuda()
    qi = torch.from_numpy(rng.integers(0, S, size=E).astype(np.int32)).cuda(); ki = torch.from_numpy(rng.integers(0, S, size=E).astype(np.int32)).cuda()
    kv = tile_planes(torch.cat((k, v), dim=1).cuda(), T, nb, npl); ldp = nb * 512 * npl
    k_ptr, v_ptr, kv_stride = kv.data_ptr(), kv.data_ptr() + 2 * D * ldp, 2 * D * ldp
    ctx = torch.zeros((E, D, N), device="cuda"); lse = torch.zeros((E, H, N), device="cuda"); sc0 = torch.zeros((E, H, nb, T, Tp), device="cuda")
    L.check(lib.csn_block_attn_fwd_f32(q.data_ptr(), k_ptr, v_ptr, D * N, kv_stride, qi.data_ptr(), ki.data_ptr(), N, ctx.data_ptr(), D * N,
                                       sc0.data_ptr(), lse.data_ptr(), E, H, d, T, nb, Tp, 8.0, 0.0, 0, 1, ldp, _stream()))
    res = {}
    for name, rc, pt in (("kept pt0", 0, 0), ("kept pt1", 0, 1), ("rc pt0", 1, 0), ("rc pt1", 1, 1)):
        sc = sc0.clone(); ds = torch.zeros_like(sc); de = torch.zeros((E, H, N), device="cuda"); dq = torch.zeros((E, D, N), device="cuda")
        if rc:
            L.check(lib.csn_block_attn_bwd_dq_recompute_f32(dctx.data_ptr(), ctx.data_ptr(), D * N, q.data_ptr(), D * N, qi.data_ptr(), k_ptr, v_ptr, kv_stride,
                                                            ki.data_ptr(), N, sc.data_ptr(), ds.data_ptr(), lse.data_ptr(), de.data_ptr(), dq.data_ptr(), D * N,
                                                            None, 0, None, E, H, d, T, nb, Tp, 0.0, 0, ldp, 0, pt, None, 0, _stream()))
        else:
            L.check(lib.csn_block_attn_bwd_dq_f32(dctx.data_ptr(), ctx.data_ptr(), D * N, k_ptr, v_ptr, kv_stride, ki.data_ptr(), N, sc.data_ptr(), ds.data_ptr(),
                                                  lse.data_ptr(), de.data_ptr(), dq.data_ptr(), D * N, None, 0, None, E, H, d, T, nb, Tp, 0.0, 0, 0, 0, 1, ldp, pt, None, 0, _stream()))
        torch.cuda.synchronize()
        res[name] = dq
    names = list(res)
    print(f"mode {mode} d {d} T {T}: dq max |diff| between calls")
    for i, a in enumerate(names):
        print("   ", a, [f"{(res[a] - res[b]).abs().max().item():.2e}" for b in names])
