#!/usr/bin/env python3
"""development aid: where do the recomputed-score dQ call and the kept-scores call differ?"""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from csn_amd import _lib as L
from tests.test_gpu_flash import tile_planes, _rand, _stream

def run(mode, S, E, H, d, T, nb, p_drop):
    lib = L.lib(); L.check(lib.csn_set_math_mode(mode))
    rng = np.random.default_rng(100 + d + T)
    D, N, npl = H * d, T * nb, (2 if mode == 1 else 1)
    Tp = (T + 31) // 32 * 32
    q = (_rand(rng, S, D, N) / math.sqrt(math.sqrt(d))).cuda()
    k = _rand(rng, S, D, N) / math.sqrt(math.sqrt(d)); v = _rand(rng, S, D, N)
    dctx = _rand(rng, E, D, N).cuda()
    q_idx = rng.integers(0, S, size=E).astype(np.int32); kv_idx = rng.integers(0, S, size=E).astype(np.int32)
    qi, ki = torch.from_numpy(q_idx).cuda(), torch.from_numpy(kv_idx).cuda()
    kv = tile_planes(torch.cat((k, v), dim=1).cuda(), T, nb, npl); ldp = nb * 512 * npl
    k_ptr, v_ptr, kv_stride = kv.data_ptr(), kv.data_ptr() + 2 * D * ldp, 2 * D * ldp
    seed = 987654321
    ctx = torch.zeros((E, D, N), device="cuda"); lse = torch.zeros((E, H, N), device="cuda")
    sc = torch.zeros((E, H, nb, T, Tp), device="cuda")
    L.check(lib.csn_block_attn_fwd_f32(q.data_ptr(), k_ptr, v_ptr, D * N, kv_stride, qi.data_ptr(), ki.data_ptr(), N, ctx.data_ptr(), D * N,
                                       sc.data_ptr(), lse.data_ptr(), E, H, d, T, nb, Tp, 8.0, p_drop, seed, 1, ldp, _stream()))
    s_keep = sc.clone()
    def bufs(): return torch.zeros((E, H, nb, T, Tp), device="cuda"), torch.zeros((E, H, N), device="cuda"), torch.zeros((E, D, N), device="cuda")
    ds0, de0, dq0 = bufs()
    L.check(lib.csn_block_attn_bwd_dq_f32(dctx.data_ptr(), ctx.data_ptr(), D * N, k_ptr, v_ptr, kv_stride, ki.data_ptr(), N, sc.data_ptr(), ds0.data_ptr(),
                                          lse.data_ptr(), de0.data_ptr(), dq0.data_ptr(), D * N, None, 0, None, E, H, d, T, nb, Tp, p_drop, seed, 0, 0, 1, ldp, 1, None, 0, _stream()))
    outs = []
    for rep in range(2):
        ds1, de1, dq1 = bufs(); pr1 = torch.zeros((E, H, nb, T, Tp), device="cuda")
        L.check(lib.csn_block_attn_bwd_dq_recompute_f32(dctx.data_ptr(), ctx.data_ptr(), D * N, q.data_ptr(), D * N, qi.data_ptr(), k_ptr, v_ptr, kv_stride,
                                                        ki.data_ptr(), N, pr1.data_ptr(), ds1.data_ptr(), lse.data_ptr(), de1.data_ptr(), dq1.data_ptr(), D * N,
                                                        None, 0, None, E, H, d, T, nb, Tp, p_drop, seed, ldp, 0, 1, None, 0, _stream()))
        outs.append((ds1, de1, dq1, pr1))
    torch.cuda.synchronize()
    ds1, de1, dq1, pr1 = outs[0]
    print(f"mode {mode} d {d} T {T} nb {nb} H {H} drop {p_drop}: rerun equal", all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])))
    print("  delta equal", torch.equal(de0, de1), " dq maxdiff", (dq0 - dq1).abs().max().item(), "of", dq0.abs().max().item(),
          " mismatching", (dq0 != dq1).float().mean().item())
    if mode == 1:
        def dec(t): 
            pl = t.view(torch.bfloat16).view(E, H, nb, T, Tp // 32, 2, 32).float(); return (pl[..., 0, :] + pl[..., 1, :]).reshape(E, H, nb, T, Tp)
        p0, p1, d0, d1 = dec(sc), dec(pr1), dec(ds0), dec(ds1)
    else:
        a = ds0.view(torch.bfloat16).view(E, H, nb, 2, T, Tp).float(); b = ds1.view(torch.bfloat16).view(E, H, nb, 2, T, Tp).float()
        p0, d0, p1, d1 = a[:, :, :, 0], a[:, :, :, 1], b[:, :, :, 0], b[:, :, :, 1]
    print("  P maxdiff", (p0 - p1).abs().max().item(), " dS maxdiff", (d0 - d1).abs().max().item(), "of", d0.abs().max().item())
    bad = (p0 != p1)
    if bad.any():
        idx = bad.nonzero()
        print("  first P mismatches (e,h,blk,q,key):", idx[:6].tolist(), " count", idx.shape[0], "keys%32:", sorted(set((idx[:, 4] % 32).tolist()))[:40],
              "tiles:", sorted(set((idx[:, 4] // 32).tolist())))
        e, h, b, qq, kk = idx[0].tolist()
        srow = s_keep[e, h, b, qq, kk].item()
        print("   S", srow, "lse", lse[e, h, b * T + qq].item(), "p0", p0[e, h, b, qq, kk].item(), "p1", p1[e, h, b, qq, kk].item())

for case in [(1, 1, 1, 1, 64, 64, 1, 0.0), (1, 3, 5, 2, 64, 100, 3, 0.0), (2, 2, 3, 1, 256, 500, 2, 0.1)]:
    run(*case)
