"""Per-entry-point parity tests of libcsn_hip.so on the MI355X: every C-ABI function against a float64 torch
restatement of the same arithmetic on the host (tolerances are fp32 rounding, far inside the 1e-4 contract)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from csn_amd import _lib
    _lib.build()
    return _lib


# every test runs in both arithmetic modes of the contractions: 0 = exact fp32 matrix cores, 1 = bf16x3 (three bf16
# products per fp32 product, ~1e-5 relative; tolerances widen by TOL_SCALE but stay inside the 1e-4 contract)
TOL_SCALE = {0: 1.0, 1: 8.0}
_mode = {"m": 0}


@pytest.fixture(autouse=True, params=[0, 1], ids=["fp32", "bf16x3"])
def math_mode(request, L):
    L.check(L.lib().csn_set_math_mode(request.param))
    _mode["m"] = request.param
    yield request.param
    L.lib().csn_set_math_mode(1)


def tol(x):
    return x * TOL_SCALE[_mode["m"]]


def _rand(rng, *shape):
    return torch.from_numpy(rng.standard_normal(size=shape).astype(np.float32))


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _maxerr(got, ref):
    ref = ref.double()
    return ((got.detach().cpu().double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("S,C,N,R,div_rows", [(2, 256, 1000, 768, 256), (3, 96, 500, 96, 0), (1, 128, 2048, 384, 128),
                                              (1, 32, 36, 32, 0),
                                              # 256 x 256 tiles with ragged edges in every direction: 200 / 460 rows (one and two
                                              # row tiles, the last one short), 236 / 1004 points, contraction lengths that end
                                              # inside a 32-wide slab, the scaled rows ending inside a tile
                                              (9, 260, 236, 200, 100), (2, 100, 1004, 460, 300), (11, 36, 520, 256, 0)])
def test_project(L, S, C, N, R, div_rows):
    from csn_amd import functional as CF
    rng = np.random.default_rng(1)
    x, w = _rand(rng, S, C, N), _rand(rng, R, C) / math.sqrt(C)
    out = CF.project(x.cuda(), w.cuda(), div_rows=div_rows, temperature=16.0)
    ref = torch.einsum("rc,scn->srn", w.double(), x.double())
    ref[:, :div_rows] /= 16.0
    assert _maxerr(out, ref) < tol(2e-6)


def test_project_split_output_planes(L, math_mode):
    """bf16x3 mode only: the projection can emit its result as bf16 hi/lo planes (x = hi + lo), see csn_hip.h."""
    from csn_amd import functional as CF
    rng = np.random.default_rng(11)
    S, C, N, R = 2, 96, 500, 192
    x, w = _rand(rng, S, C, N), _rand(rng, R, C) / math.sqrt(C)
    if math_mode == 0:
        out = torch.empty((S, 2, R, N), device="cuda", dtype=torch.bfloat16)
        xd, wd = x.cuda(), w.cuda()
        rc = L.lib().csn_project_f32(xd.data_ptr(), C * N, N, wd.data_ptr(), R, C, out.data_ptr(), 2 * R * N, N, S, N, 0, 1.0,
                                     1, R * N, _stream())
        assert rc == -1                                    # split tensors exist only in the bf16x3 mode
        return
    planes = CF.project(x.cuda(), w.cuda(), split=True)    # (S, 2, R, N) bf16
    assert planes.dtype == torch.bfloat16 and planes.shape == (S, 2, R, N)
    ref = torch.einsum("rc,scn->srn", w.double(), x.double())
    got = planes[:, 0].double().cpu() + planes[:, 1].double().cpu()
    assert ((got - ref).abs().max() / ref.abs().max()).item() < 3e-5
    hi = planes[:, 0].float().cpu()
    assert ((hi.double() - ref).abs().max() / ref.abs().max()).item() < 5e-3      # the high plane alone is bf16-accurate


@pytest.mark.parametrize("S,C,R,T,nb", [(2, 96, 192, 100, 3), (1, 256, 512, 500, 2), (1, 32, 64, 36, 2)])
def test_project_tile_planes_feed_attention(L, math_mode, S, C, R, T, nb):
    """bf16x3 mode: csn_project_f32(out_split = 2) writes K/V as tile planes — per row and block, 16 tiles of
    [hi 32 | lo 32] bf16 at block pitch 1024 — and the attention forward fed with them agrees with the fp32-fed one."""
    if math_mode == 0:
        return
    rng = np.random.default_rng(21)
    N = T * nb
    x, w = _rand(rng, S, C, N), _rand(rng, R, C) / math.sqrt(C)
    xd, wd = x.cuda(), w.cuda()
    ldp = nb * 1024
    kv = torch.full((S, R, ldp), float("nan"), device="cuda", dtype=torch.bfloat16)
    L.check(L.lib().csn_project_f32(xd.data_ptr(), C * N, N, wd.data_ptr(), R, C, kv.data_ptr(), R * ldp, ldp, S, N, 0, 1.0, 2, T,
                                    _stream()))
    t = kv.view(S, R, nb, 16, 2, 32).float().cpu()
    got = (t[..., 0, :] + t[..., 1, :]).reshape(S, R, nb, 512)
    ref = torch.einsum("rc,scn->srn", w.double(), x.double()).view(S, R, nb, T)
    assert ((got[..., :T].double() - ref).abs().max() / ref.abs().max()).item() < 3e-5
    last = (T + 31) // 32 * 32
    assert (got[..., T:last] == 0).all()                         # the padding keys of the last tile are written as zeros
    assert torch.isnan(got[..., last:]).all()                    # tiles beyond it are never touched (and never read)
    # attention forward: K = rows [0, R/2), V = rows [R/2, R) as tile planes vs the same values as fp32 maps
    d = R // 2
    if d not in (32, 64, 96, 128, 256):
        return
    q = (_rand(rng, S, d, N) / math.sqrt(d)).cuda()              # logits of O(1): operand rounding is not amplified by exp
    kvf = torch.einsum("rc,scn->srn", wd, xd).contiguous()       # fp32 (torch) K|V maps: same values up to fp32 rounding
    Tp = (T + 31) // 32 * 32
    outs = []
    for tiles in (0, 1):
        ctx = torch.empty((S, d, N), device="cuda")
        lse = torch.empty((S, 1, N), device="cuda")
        kp = kv.data_ptr() if tiles else kvf.data_ptr()
        vp = kp + (2 * d * ldp if tiles else 4 * d * N)
        L.check(L.lib().csn_block_attn_fwd_f32(q.data_ptr(), kp, vp, d * N, (R * ldp) if tiles else (R * N), None, None, N,
                                               ctx.data_ptr(), d * N, None, lse.data_ptr(), S, 1, d, T, nb, Tp, 8.0, 0.0, 0,
                                               tiles, ldp if tiles else 0, _stream()))
        outs.append((ctx, lse))
    assert _maxerr(outs[1][0], outs[0][0].cpu()) < 1e-4 and _maxerr(outs[1][1], outs[0][1].cpu()) < 1e-4


def test_project_wgrad(L):
    from csn_amd import functional as CF
    rng = np.random.default_rng(2)
    for (S, R, C, N, NP) in [(3, 768, 256, 1000, 1000), (2, 96, 96, 5000, 4800), (1, 256, 128, 260, 260)]:
        dout, x = _rand(rng, S, R, NP), _rand(rng, S, C, N)
        dw = CF.project_wgrad(dout.cuda(), x.cuda(), scale=0.5)
        ref = 0.5 * torch.einsum("srn,scn->rc", dout.double(), x[:, :, :NP].double())
        assert _maxerr(dw, ref) < tol(5e-6), (S, R, C, N)


# ---------------------------------------------------------------------------------------------------------
def _attn_reference(q, k, v, q_idx, kv_idx, H, d, T, nb):
    """q,k,v: (S, H*d, N) float64 channel-major (q already scaled). Returns ctx (E,H*d,NP), lse (E,H,NP), scores [query][key]."""
    E = len(q_idx)
    NP = T * nb
    ctx = torch.zeros(E, H * d, NP, dtype=torch.float64)
    lse = torch.zeros(E, H, NP, dtype=torch.float64)
    sc = torch.zeros(E, H, nb, T, T, dtype=torch.float64)
    for e in range(E):
        for h in range(H):
            for b in range(nb):
                sl = slice(b * T, (b + 1) * T)
                qq = q[q_idx[e], h * d:(h + 1) * d, sl]          # (d, T)
                kk = k[kv_idx[e], h * d:(h + 1) * d, sl]
                vv = v[kv_idx[e], h * d:(h + 1) * d, sl]
                s = qq.t() @ kk                                   # (Tq, Tk)
                p = torch.softmax(s, dim=-1)
                ctx[e, h * d:(h + 1) * d, sl] = (p @ vv.t()).t()
                lse[e, h, sl] = torch.logsumexp(s, dim=-1)
                sc[e, h, b] = s                                   # [query][key]
    return ctx, lse, sc


ATTN_CASES = [
    # S, E, H, d, T, nb
    (3, 5, 2, 64, 100, 3),
    (2, 3, 1, 256, 500, 2),
    (2, 2, 1, 128, 512, 1),
    (1, 1, 2, 96, 36, 2),
    (2, 2, 1, 32, 132, 1),
]


def _attn_inputs(rng, S, E, H, d, T, nb, extra_ld=0):
    N = T * nb + extra_ld
    q = _rand(rng, S, H * d, N) / math.sqrt(math.sqrt(d))
    k = _rand(rng, S, H * d, N) / math.sqrt(math.sqrt(d))
    v = _rand(rng, S, H * d, N)
    q_idx = rng.integers(0, S, size=E).astype(np.int32)
    kv_idx = rng.integers(0, S, size=E).astype(np.int32)
    return q, k, v, q_idx, kv_idx


def _run_attn_fwd(L, q, k, v, q_idx, kv_idx, H, d, T, nb, thr=8.0, keep=True, p_drop=0.0, seed=0):
    S, D, N = q.shape
    E = len(q_idx)
    Tp = (T + 31) // 32 * 32
    dev = "cuda"
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    qi, ki = torch.from_numpy(q_idx).cuda(), torch.from_numpy(kv_idx).cuda()
    ctx = torch.full((E, D, N), float("nan"), device=dev)
    lse = torch.zeros((E, H, T * nb), device=dev)
    scores = torch.full((E, H, nb, T, Tp), float("nan"), device=dev) if keep else None
    rc = L.lib().csn_block_attn_fwd_f32(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), D * N, D * N, qi.data_ptr(),
                                        ki.data_ptr(), N, ctx.data_ptr(), D * N, scores.data_ptr() if keep else None,
                                        lse.data_ptr(), E, H, d, T, nb, Tp, thr, p_drop, seed, 0, 0, _stream())
    L.check(rc, "attn fwd")
    torch.cuda.synchronize()
    return ctx, lse, scores, (qd, kd, vd, qi, ki)


@pytest.mark.parametrize("S,E,H,d,T,nb", ATTN_CASES)
def test_block_attn_fwd(L, S, E, H, d, T, nb):
    rng = np.random.default_rng(3)
    q, k, v, q_idx, kv_idx = _attn_inputs(rng, S, E, H, d, T, nb, extra_ld=8)
    ctx, lse, scores, _ = _run_attn_fwd(L, q, k, v, q_idx, kv_idx, H, d, T, nb)
    NP = T * nb
    rctx, rlse, rsc = _attn_reference(q.double(), k.double(), v.double(), q_idx, kv_idx, H, d, T, nb)
    assert _maxerr(ctx[:, :, :NP], rctx) < tol(5e-6)
    assert torch.isnan(ctx[:, :, NP:]).all()                    # columns past n_blocks*block are never written
    assert (lse.cpu().double() - rlse).abs().max().item() < tol(1e-5)
    assert (scores[..., :T].cpu().double() - rsc).abs().max().item() < tol(1e-5)
    assert torch.isnan(scores[..., T:]).all()                   # padding of the score rows is never written


def test_block_attn_fwd_rescale_branch(L):
    """The lazy softmax re-basing (threshold 8) must agree with re-basing on every tile (threshold 0), also when one
    late key dominates a query row by far (forces the rare branch: cdna guide rule 26)."""
    rng = np.random.default_rng(4)
    S, E, H, d, T, nb = 1, 1, 1, 64, 200, 1
    q, k, v, q_idx, kv_idx = _attn_inputs(rng, S, E, H, d, T, nb)
    q_idx[:] = 0
    kv_idx[:] = 0
    k[0, :, 150] = q[0, :, 17] * 40.0          # key 150 (5th key tile) spikes for query 17
    k[0, :, 3] = q[0, :, 90] * 25.0            # key 3 (first tile) spikes for query 90
    outs = [_run_attn_fwd(L, q, k, v, q_idx, kv_idx, H, d, T, nb, thr=t)[:2] for t in (0.0, 8.0, 30.0)]
    rctx, rlse, _ = _attn_reference(q.double(), k.double(), v.double(), q_idx, kv_idx, H, d, T, nb)
    # the spiked logits reach |s| ~ 300: an input-relative error of 1e-5 (bf16x3) is 3e-3 absolute in the exponent there
    lim = 5e-6 if _mode["m"] == 0 else 2e-4
    for ctx, lse in outs:
        assert _maxerr(ctx, rctx) < lim
        assert (lse.cpu().double() - rlse).abs().max().item() < tol(2e-5) * rlse.abs().max().item()


@pytest.mark.parametrize("S,E,H,d,T,nb", ATTN_CASES)
def test_block_attn_bwd(L, S, E, H, d, T, nb):
    rng = np.random.default_rng(5)
    q, k, v, q_idx, kv_idx = _attn_inputs(rng, S, E, H, d, T, nb)
    ctx, lse, scores, (qd, kd, vd, qi, ki) = _run_attn_fwd(L, q, k, v, q_idx, kv_idx, H, d, T, nb)
    D, N = H * d, T * nb
    Tp = scores.shape[-1]
    dctx = _rand(rng, E, D, N)
    dd = dctx.cuda()
    dscores = torch.full_like(scores, float("nan"))
    delta = torch.empty((E, H, N), device="cuda")
    dq, dk, dv = (torch.full((E, D, N), float("nan"), device="cuda") for _ in range(3))
    pt = 1 if _mode["m"] == 1 else 0          # bf16x3: P / dS travel from the dq call to the dkv call as bf16 tile planes
    # per-evaluation outputs: identity slot maps, no accumulation (the module path uses slot maps + colours)
    rc = L.lib().csn_block_attn_bwd_dq_f32(dd.data_ptr(), ctx.data_ptr(), D * N, kd.data_ptr(), vd.data_ptr(), D * N,
                                           ki.data_ptr(), N, scores.data_ptr(), dscores.data_ptr(), lse.data_ptr(),
                                           delta.data_ptr(), dq.data_ptr(), D * N, None, 0, None, E, H, d, T, nb, Tp, 0.0, 0,
                                           0, 0, 0, 0, pt, None, 0, _stream())
    L.check(rc, "attn bwd dq")
    rc = L.lib().csn_block_attn_bwd_dkv_f32(dd.data_ptr(), D * N, qd.data_ptr(), D * N, qi.data_ptr(), N, scores.data_ptr(),
                                            dscores.data_ptr(), dk.data_ptr(), dv.data_ptr(), D * N, None, None, 0, None, E,
                                            H, d, T, nb, Tp, 0, 0, 0, 0, pt, None, 0, _stream())
    L.check(rc, "attn bwd")
    torch.cuda.synchronize()
    # float64 autograd reference, per evaluation (no sharing: the ABI returns per-evaluation gradients)
    q64 = q.double()[q_idx].clone().requires_grad_(True)
    k64 = k.double()[kv_idx].clone().requires_grad_(True)
    v64 = v.double()[kv_idx].clone().requires_grad_(True)
    ident = np.arange(E)
    rctx, _, _ = _attn_reference_autograd(q64, k64, v64, H, d, T, nb)
    rctx.backward(dctx.double())
    assert _maxerr(dq, q64.grad) < tol(2e-5)
    assert _maxerr(dk, k64.grad) < tol(2e-5)
    assert _maxerr(dv, v64.grad) < tol(2e-5)
    # scores now hold P
    p_ref = torch.softmax(_attn_reference(q.double(), k.double(), v.double(), q_idx, kv_idx, H, d, T, nb)[2], dim=-1)
    if pt:      # per query row: tiles of [hi 32 | lo 32] bf16 over the bytes of the fp32 row
        pl = scores.view(torch.bfloat16).view(E, H, nb, T, Tp // 32, 2, 32).float()
        got = (pl[..., 0, :] + pl[..., 1, :]).reshape(E, H, nb, T, Tp)
    else:
        got = scores
    assert (got[..., :T].cpu().double() - p_ref).abs().max().item() < tol(2e-6)
    if pt:
        assert (got[..., T:] == 0).all()                          # keys beyond the block end are written as zeros

    # slot-indexed accumulation: evaluations listed in two disjoint colours, gradients summed per slot
    qi64, ki64 = torch.from_numpy(q_idx).long(), torch.from_numpy(kv_idx).long()
    ref_dq = torch.zeros(S, D, N, dtype=torch.float64).index_add_(0, qi64, q64.grad)
    ref_dk = torch.zeros(S, D, N, dtype=torch.float64).index_add_(0, ki64, k64.grad)
    ref_dv = torch.zeros(S, D, N, dtype=torch.float64).index_add_(0, ki64, v64.grad)
    from csn_amd.functional import EvalPlan
    plan = EvalPlan(q_idx, kv_idx, S, "cuda")
    _, _, scores2, _ = _run_attn_fwd(L, q, k, v, q_idx, kv_idx, H, d, T, nb)
    sq, sk, sv_ = (torch.zeros((S, D, N), device="cuda") for _ in range(3))
    for ids in plan.dq_colors:
        L.check(L.lib().csn_block_attn_bwd_dq_f32(dd.data_ptr(), ctx.data_ptr(), D * N, kd.data_ptr(), vd.data_ptr(), D * N,
                                                  ki.data_ptr(), N, scores2.data_ptr(), dscores.data_ptr(), lse.data_ptr(),
                                                  delta.data_ptr(), sq.data_ptr(), D * N, qi.data_ptr(), 1, ids.data_ptr(),
                                                  ids.numel(), H, d, T, nb, Tp, 0.0, 0, 0, 0, 0, 0, pt, None, 0, _stream()))
    for ids in plan.dkv_colors:
        L.check(L.lib().csn_block_attn_bwd_dkv_f32(dd.data_ptr(), D * N, qd.data_ptr(), D * N, qi.data_ptr(), N,
                                                   scores2.data_ptr(), dscores.data_ptr(), sk.data_ptr(), sv_.data_ptr(), D * N,
                                                   ki.data_ptr(), ki.data_ptr(), 1, ids.data_ptr(), ids.numel(), H, d, T, nb,
                                                   Tp, 0, 0, 0, 0, pt, None, 0, _stream()))
    torch.cuda.synchronize()
    assert _maxerr(sq, ref_dq) < tol(2e-5) and _maxerr(sk, ref_dk) < tol(2e-5) and _maxerr(sv_, ref_dv) < tol(2e-5)
    # grouped form of the dK / dV call: the evaluations of a slot are accumulated in registers, every slot is written once
    grouping = L.lib().csn_attn_bwd_grouping(d, T)
    assert (grouping & 1) == _mode["m"]
    if grouping & 1:
        gq = torch.full((S, D, N), float("nan"), device="cuda")
        _, _, scores2, _ = _run_attn_fwd(L, q, k, v, q_idx, kv_idx, H, d, T, nb)       # fresh scores: the dq call consumes them
        dscores.fill_(float("nan"))
        L.check(L.lib().csn_block_attn_bwd_dq_f32(dd.data_ptr(), ctx.data_ptr(), D * N, kd.data_ptr(), vd.data_ptr(), D * N,
                                                  ki.data_ptr(), N, scores2.data_ptr(), dscores.data_ptr(), lse.data_ptr(),
                                                  delta.data_ptr(), gq.data_ptr(), D * N, qi.data_ptr(), 0,
                                                  plan.q_group_items.data_ptr(), E, H, d, T, nb, Tp, 0.0, 0, 0, 0, 0, 0, pt,
                                                  plan.q_group_off.data_ptr(), plan.n_q_groups, _stream()))
        torch.cuda.synchronize()
        used_q = torch.from_numpy(np.unique(q_idx)).long()
        assert _maxerr(gq[used_q], ref_dq[used_q]) < tol(2e-5)
    if grouping & 2:
        gk, gv = (torch.full((S, D, N), float("nan"), device="cuda") for _ in range(2))
        L.check(L.lib().csn_block_attn_bwd_dkv_f32(dd.data_ptr(), D * N, qd.data_ptr(), D * N, qi.data_ptr(), N,
                                                   scores2.data_ptr(), dscores.data_ptr(), gk.data_ptr(), gv.data_ptr(), D * N,
                                                   ki.data_ptr(), ki.data_ptr(), 0, plan.kv_group_items.data_ptr(), E, H, d, T,
                                                   nb, Tp, 0, 0, 0, 0, pt, plan.kv_group_off.data_ptr(), plan.n_kv_groups,
                                                   _stream()))
        torch.cuda.synchronize()
        used = torch.from_numpy(np.unique(kv_idx)).long()
        assert _maxerr(gk[used], ref_dk[used]) < tol(2e-5) and _maxerr(gv[used], ref_dv[used]) < tol(2e-5)
    else:
        assert _mode["m"] == 0 or d < 192 or T < 224


def _attn_reference_autograd(q, k, v, H, d, T, nb):
    E = q.shape[0]
    qb = q.view(E, H, d, nb, T).permute(0, 1, 3, 4, 2)          # (E,H,nb,T,d)
    kb = k.view(E, H, d, nb, T).permute(0, 1, 3, 4, 2)
    vb = v.view(E, H, d, nb, T).permute(0, 1, 3, 4, 2)
    p = torch.softmax(qb @ kb.transpose(-1, -2), dim=-1)
    o = p @ vb                                                  # (E,H,nb,T,d)
    return o.permute(0, 1, 4, 2, 3).reshape(E, H * d, nb * T), None, None


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("S,E,C,D,NP", [(2, 3, 256, 256, 1000), (1, 2, 96, 192, 500), (2, 2, 128, 128, 2048), (1, 1, 32, 64, 36)])
def test_outproj_ln_fwd_bwd(L, S, E, C, D, NP):
    rng = np.random.default_rng(6)
    att = _rand(rng, E, D, NP)
    wfc = _rand(rng, C, D) / math.sqrt(D)
    x = _rand(rng, S, C, NP)
    ridx = rng.integers(0, S, size=E).astype(np.int32)
    dxhat = _rand(rng, E, C, NP)
    dev = "cuda"
    attd, wd, xd, rid = att.cuda(), wfc.cuda(), x.cuda(), torch.from_numpy(ridx).cuda()
    xhat = torch.empty((E, C, NP), device=dev)
    rstd = torch.empty((E, NP), device=dev)
    # point sums of xhat (the pooled descriptors): with a workspace (fused into the epilogue where the kernel can) and without
    sums = torch.full((E, C), float("nan"), device=dev)
    sums_nows = torch.full((E, C), float("nan"), device=dev)
    ws_n = E * ((NP + 255) // 256) * C
    ws = torch.empty((ws_n,), device=dev)
    L.check(L.lib().csn_outproj_ln_fwd_f32(attd.data_ptr(), D * NP, wd.data_ptr(), xd.data_ptr(), C * NP, rid.data_ptr(),
                                           xhat.data_ptr(), C * NP, rstd.data_ptr(), E, C, D, NP, NP, 1e-6, 0.0, 0,
                                           sums_nows.data_ptr(), None, 0, _stream()))
    xhat.fill_(float("nan"))
    L.check(L.lib().csn_outproj_ln_fwd_f32(attd.data_ptr(), D * NP, wd.data_ptr(), xd.data_ptr(), C * NP, rid.data_ptr(),
                                           xhat.data_ptr(), C * NP, rstd.data_ptr(), E, C, D, NP, NP, 1e-6, 0.0, 0,
                                           sums.data_ptr(), ws.data_ptr(), ws_n, _stream()))
    a64 = att.double().requires_grad_(True)
    w64 = wfc.double().requires_grad_(True)
    z = torch.einsum("cd,edn->ecn", w64, a64) + x.double()[ridx]
    mean = z.mean(dim=1, keepdim=True)
    var = z.var(dim=1, unbiased=False, keepdim=True)
    ref = (z - mean) / torch.sqrt(var + 1e-6)
    assert _maxerr(xhat, ref) < tol(5e-6)
    assert _maxerr(rstd, (1 / torch.sqrt(var + 1e-6)).squeeze(1)) < tol(5e-6)
    ref_sums = ref.sum(dim=2)                                           # O(sqrt(NP)) numbers summing O(1) terms
    for got in (sums, sums_nows):
        assert ((got.cpu().double() - ref_sums).abs().max() / NP).item() < tol(2e-6)

    dz = torch.empty((E, C, NP), device=dev)
    datt = torch.empty((E, D, NP), device=dev)
    dw = torch.empty((C, D), device=dev)
    ws_n = L.lib().csn_wgrad_workspace_floats(C, D, E, NP)
    ws = torch.empty((ws_n,), device=dev)
    wt = wd.t().contiguous()
    dxd = dxhat.cuda()
    # incoming gradient = dense maps for the first E-1 evaluations only + a per-row constant (the pooled-mean term)
    rows = _rand(rng, E, C)
    rows_d = rows.cuda()
    nd = E - 1
    L.check(L.lib().csn_outproj_ln_bwd_f32(dxd.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), C * NP, attd.data_ptr(), D * NP,
                                           wt.data_ptr(), dz.data_ptr(), None, datt.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws_n,
                                           E, C, D, NP, NP, 0, 0.0, 0, 0, 0, rows_d.data_ptr(), nd, None, 1, _stream()))
    dx_eff = dxhat.double().clone()
    dx_eff[nd:] = 0
    dx_eff += rows.double()[:, :, None]
    ref.backward(dx_eff, retain_graph=True)
    assert _maxerr(datt, a64.grad) < tol(2e-5)
    assert _maxerr(dw, w64.grad) < tol(2e-5)
    # scaled / grouped dense term: evaluation e < nd takes scale[e][c] * dxhat[e // 2][c][n] (the mix gradient rebuilt on the fly)
    scale = _rand(rng, E, C)
    L.check(L.lib().csn_outproj_ln_bwd_f32(dxd.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), C * NP, attd.data_ptr(), D * NP,
                                           wt.data_ptr(), dz.data_ptr(), None, datt.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws_n,
                                           E, C, D, NP, NP, 0, 0.0, 0, 0, 0, rows_d.data_ptr(), nd, scale.cuda().data_ptr(), 2,
                                           _stream()))
    dx_eff = torch.zeros_like(dx_eff)
    for e in range(nd):
        dx_eff[e] = scale[e].double()[:, None] * dxhat[e // 2].double()
    dx_eff += rows.double()[:, :, None]
    a64.grad = None
    w64.grad = None
    ref.backward(dx_eff)
    assert _maxerr(datt, a64.grad) < tol(2e-5)
    assert _maxerr(dw, w64.grad) < tol(2e-5)


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("split", [False, True], ids=["one-tensor", "self-separate"])
@pytest.mark.parametrize("B,K1,C,NP", [(3, 4, 24, 100), (2, 1, 16, 36), (5, 3, 40, 1000)])
def test_mix_forward_backward(L, B, K1, C, NP, split):
    """csn_mix_fwd_f32 / csn_mix_bwd_f32: feats[b] = sum_k comp[b,k] (gamma xhat[b,k] + beta), the k = 0 maps either in the
    same tensor or (overlapped multi-GPU path) in their own."""
    from csn_amd import functional as CF
    if split and K1 == 1:
        pytest.skip("k1 = 1 has no second tensor")
    rng = np.random.default_rng(B * 100 + K1)
    xh = _rand(rng, B, K1, C, NP)
    comp = torch.softmax(_rand(rng, B, K1), dim=1)
    gamma, beta = _rand(rng, C), _rand(rng, C)
    dfe = _rand(rng, B, C, NP)
    x64, c64, g64, b64 = (t.double().requires_grad_() for t in (xh, comp, gamma, beta))
    ref = (c64[:, :, None, None] * (x64 * g64[None, None, :, None] + b64[None, None, :, None])).sum(dim=1)
    ref.backward(dfe.double())
    cg, gg, bg = (t.cuda().requires_grad_() for t in (comp, gamma, beta))
    if split:
        own = xh[:, 0].contiguous().cuda().requires_grad_()
        rest = xh[:, 1:].reshape(B * (K1 - 1), C, NP).contiguous().cuda().requires_grad_()
        got = CF.csa_mix(rest, cg, gg, bg, B, K1, xself=own)
    else:
        allm = xh.reshape(B * K1, C, NP).cuda().requires_grad_()
        got = CF.csa_mix(allm, cg, gg, bg, B, K1)
    got.backward(dfe.cuda())
    assert _maxerr(got, ref) < 1e-5
    if split:
        dx = torch.cat((own.grad.view(B, 1, C, NP), rest.grad.view(B, K1 - 1, C, NP)), dim=1)
    else:
        dx = allm.grad.view(B, K1, C, NP)
    assert _maxerr(dx, x64.grad) < 1e-5
    assert _maxerr(cg.grad, c64.grad) < 1e-4 * max(1.0, c64.grad.abs().max().item())
    assert _maxerr(gg.grad, g64.grad) < 1e-4 * max(1.0, g64.grad.abs().max().item())
    assert _maxerr(bg.grad, b64.grad) < 1e-4 * max(1.0, b64.grad.abs().max().item())


# ---------------------------------------------------------------------------------------------------------
def test_retrieval_measure_against_oracle(L):
    from csn_amd import functional as CF
    from oracle import csa_oracle as orc
    rng = np.random.default_rng(7)
    f1 = orc.synth_clustered_feats(rng, 3, 200)
    f2 = orc.synth_clustered_feats(rng, 5, 333)
    got = CF.retrieval_measure(f1.cuda(), f2.cuda()).cpu()
    ref = orc.retrieval_measure(f1, f2)
    assert (got - ref).abs().max().item() < tol(2e-6)


def test_retrieval_measure_many_pairs_and_row_chunks(L):
    """More than 65536 (query, candidate) pairs in one call (the pair index lives on grid.x), and the row-chunked form that
    bounds the O(S^2) scratch: bitwise the same scores whatever the chunking, and equal to the closed form."""
    from csn_amd import functional as CF
    rng = np.random.default_rng(8)
    S1, S2, N, C = 300, 260, 8, 32
    f1 = torch.from_numpy(rng.standard_normal((S1, N, C)).astype(np.float32)).cuda()
    f2 = torch.from_numpy(rng.standard_normal((S2, N, C)).astype(np.float32)).cuda()
    whole = CF.retrieval_measure(f1, f2)
    chunked = CF.retrieval_measure(f1, f2, pair_budget=37 * S2 * N)          # 37 query shapes per call: 9 calls
    assert torch.equal(whole, chunked)
    n1 = torch.nn.functional.normalize(f1.double(), dim=-1, eps=1e-12)
    n2 = torch.nn.functional.normalize(f2.double(), dim=-1, eps=1e-12)
    ref = torch.einsum("inc,jmc->ijnm", n1, n2).amax(dim=-1).mean(dim=-1)
    assert (whole.double() - ref).abs().max().item() < 2e-6


def test_abi_rejects_bad_arguments(L):
    lib = L.lib()
    x = torch.zeros(1, 32, 36, device="cuda")
    w = torch.zeros(32, 32, device="cuda")
    out = torch.zeros(1, 32, 36, device="cuda")
    # leading dimension not a multiple of 4
    rc = lib.csn_project_f32(x.data_ptr(), 32 * 35, 35, w.data_ptr(), 32, 32, out.data_ptr(), 32 * 36, 36, 1, 35, 0, 1.0, 0, 0, _stream())
    assert rc == -2
    assert b"multiple of 4" in lib.csn_status_string(rc)
    rc = lib.csn_project_f32(None, 0, 36, w.data_ptr(), 32, 32, out.data_ptr(), 32 * 36, 36, 1, 36, 0, 1.0, 0, 0, _stream())
    assert rc == -1
    rc = lib.csn_block_attn_fwd_f32(x.data_ptr(), x.data_ptr(), x.data_ptr(), 0, 0, None, None, 36, out.data_ptr(), 0, None,
                                    None, 1, 1, 48, 36, 1, 64, 8.0, 0.0, 0, 0, 0, _stream())
    assert rc == -5


def _mutations(lib_fn, names, valid, cases):
    """valid: dict name -> value (a call that succeeds).  cases: [(name, bad value, expected status or None = any error)]."""
    call = lambda kw: lib_fn(*[kw[n] for n in names])
    assert call(valid) == 0, "the unmutated call must succeed"
    torch.cuda.synchronize()
    for name, bad, want in cases:
        rc = call({**valid, name: bad})
        assert rc < 0, f"{name}={bad!r} was accepted"
        if want is not None:
            assert rc == want, f"{name}={bad!r}: status {rc}, expected {want}"


def test_abi_argument_validation_table(L):
    """Every entry point of the hot path refuses, with the documented status and BEFORE any launch, arguments its kernels cannot
    take: null / misaligned pointers, sizes and strides that are not multiples of 4, head / model widths without a kernel
    instance, non-positive counts, dropout rates outside [0, 1), workspaces that are too small."""
    lib = L.lib()
    L.check(lib.csn_set_math_mode(0))
    try:
        ARG, ALIGN, PTR, STRIDE, DIM, WS = -1, -2, -3, -4, -5, -6
        dev = "cuda"
        S, C, N, R = 2, 64, 40, 64
        x = torch.randn(S, C, N, device=dev); w = torch.randn(R, C, device=dev); out = torch.empty(S, R, N, device=dev)
        names = ["x", "xs", "ldx", "w", "rows", "ch", "out", "os", "ldo", "ns", "np", "div", "temp", "split", "ps", "st"]
        valid = dict(x=x.data_ptr(), xs=C * N, ldx=N, w=w.data_ptr(), rows=R, ch=C, out=out.data_ptr(), os=R * N, ldo=N, ns=S,
                     np=N, div=0, temp=1.0, split=0, ps=0, st=_stream())
        _mutations(lib.csn_project_f32, names, valid, [
            ("x", None, ARG), ("w", None, ARG), ("out", None, ARG), ("x", x.data_ptr() + 4, PTR), ("out", out.data_ptr() + 8, PTR),
            ("ldx", N + 1, ALIGN), ("ldo", N + 2, ALIGN), ("np", N - 2, ALIGN), ("ldx", N - 4, ARG), ("ldo", N - 4, ARG), ("xs", C * N + 1, STRIDE), ("os", R * N + 2, STRIDE),
            ("ns", 0, ARG), ("rows", 0, ARG), ("ch", -4, ARG), ("np", 0, ARG), ("np", N + 4, ARG), ("ch", 62, ALIGN)])

        # block attention forward: 1 evaluation, 2 heads of 32, 2 blocks of 20
        H, d, T, nb, Tp = 2, 32, 20, 2, 32
        q = torch.randn(1, H * d, N, device=dev); ctx = torch.empty(1, H * d, N, device=dev)
        lse = torch.empty(1, H, T * nb, device=dev); sc = torch.empty(1, H, nb, T, Tp, device=dev)
        names = ["q", "k", "v", "qs", "kvs", "qi", "kvi", "ld", "ctx", "cs", "sc", "lse", "E", "H", "d", "T", "nb", "Tp", "thr", "p",
                 "seed", "split", "ps", "st"]
        valid = dict(q=q.data_ptr(), k=q.data_ptr(), v=q.data_ptr(), qs=H * d * N, kvs=H * d * N, qi=None, kvi=None, ld=N,
                     ctx=ctx.data_ptr(), cs=H * d * N, sc=sc.data_ptr(), lse=lse.data_ptr(), E=1, H=H, d=d, T=T, nb=nb, Tp=Tp,
                     thr=8.0, p=0.0, seed=0, split=0, ps=0, st=_stream())
        _mutations(lib.csn_block_attn_fwd_f32, names, valid, [
            ("q", None, ARG), ("ctx", None, ARG), ("d", 48, DIM), ("d", 0, None), ("ld", N - 1, ALIGN),
            ("Tp", 30, ALIGN), ("Tp", 16, ALIGN), ("q", q.data_ptr() + 4, PTR), ("sc", sc.data_ptr() + 4, PTR),
            ("qs", H * d * N + 2, STRIDE), ("cs", H * d * N + 1, STRIDE), ("E", 0, ARG), ("H", 0, ARG), ("T", 0, ARG), ("nb", 0, ARG),
            ("p", 1.0, ARG), ("p", -0.1, ARG), ("nb", 3, ARG)])          # 3 blocks of 20 > 40 points: the last block would be empty
        # the same forward grouped by query slot (round 5): the lists are device arrays, checked for presence and count only
        ids = torch.zeros(1, device=dev, dtype=torch.int32); off = torch.tensor([0, 1], device=dev, dtype=torch.int32)
        _mutations(lib.csn_block_attn_fwd_grouped_f32, names[:-1] + ["ids", "off", "ng", "st"],
                   {**valid, "ids": ids.data_ptr(), "off": off.data_ptr(), "ng": 1}, [
            ("ids", None, ARG), ("off", None, ARG), ("ng", 0, ARG), ("ng", 2, ARG), ("q", None, ARG), ("d", 48, DIM), ("ld", N - 1, ALIGN)])

        # cross-length and ragged-batch forward
        lq, lk = 24, 37
        kx = torch.randn(1, H * d, 40, device=dev)
        qx = torch.randn(1, H * d, lq, device=dev); cx = torch.empty(1, H * d, lq, device=dev)
        lse2 = torch.empty(1, H, lq, device=dev); sc2 = torch.empty(1, H, lq, 64, device=dev)
        names = ["q", "k", "v", "qs", "kvs", "ldq", "ldkv", "ctx", "cs", "sc", "lse", "E", "H", "d", "nq", "nk", "Tp", "thr", "p", "seed", "st"]
        valid = dict(q=qx.data_ptr(), k=kx.data_ptr(), v=kx.data_ptr(), qs=H * d * lq, kvs=H * d * 40, ldq=lq, ldkv=40,
                     ctx=cx.data_ptr(), cs=H * d * lq, sc=sc2.data_ptr(), lse=lse2.data_ptr(), E=1, H=H, d=d, nq=lq, nk=lk, Tp=64,
                     thr=8.0, p=0.0, seed=0, st=_stream())
        _mutations(lib.csn_cross_attn_fwd_f32, names, valid, [
            ("q", None, ARG), ("k", None, ARG), ("d", 40, DIM), ("nq", 22, ALIGN), ("nq", 0, ARG), ("nk", 0, ARG),
            ("nk", 41, ARG), ("nq", 28, ARG), ("Tp", 36, None), ("ldq", 23, ALIGN), ("p", 1.5, ARG), ("k", kx.data_ptr() + 4, PTR)])
        nql = torch.tensor([lq], dtype=torch.int32, device=dev); nkl = torch.tensor([lk], dtype=torch.int32, device=dev)
        names = ["q", "k", "v", "qs", "kvs", "ldq", "ldkv", "ctx", "cs", "sc", "lse", "E", "H", "d", "mq", "mk", "nq", "nk", "Tp", "thr", "p",
                 "seed", "st"]
        valid = dict(q=qx.data_ptr(), k=kx.data_ptr(), v=kx.data_ptr(), qs=H * d * lq, kvs=H * d * 40, ldq=lq, ldkv=40,
                     ctx=cx.data_ptr(), cs=H * d * lq, sc=sc2.data_ptr(), lse=lse2.data_ptr(), E=1, H=H, d=d, mq=lq, mk=lk,
                     nq=nql.data_ptr(), nk=nkl.data_ptr(), Tp=64, thr=8.0, p=0.0, seed=0, st=_stream())
        _mutations(lib.csn_varlen_attn_fwd_f32, names, valid, [
            ("nq", None, ARG), ("nk", None, ARG), ("mq", 22, ALIGN), ("mq", 0, ARG), ("mk", 41, ARG), ("d", 16, DIM), ("q", None, ARG)])

        # cross-length backward (scores / lse / ctx of the forward above)
        dcx = torch.randn(1, H * d, lq, device=dev); dsc = torch.empty_like(sc2); dl = torch.empty(1, H, lq, device=dev)
        dq_ = torch.empty(1, H * d, lq, device=dev); dk_ = torch.empty(1, H * d, 40, device=dev); dv_ = torch.empty(1, H * d, 40, device=dev)
        work = sc2.clone()
        names = ["dctx", "ctx", "cs", "q", "k", "v", "qs", "kvs", "ldq", "ldkv", "sc", "dsc", "lse", "delta", "dq", "dk", "dv", "dqs", "dkvs",
                 "E", "H", "d", "nq", "nk", "Tp", "p", "seed", "st"]
        valid = dict(dctx=dcx.data_ptr(), ctx=cx.data_ptr(), cs=H * d * lq, q=qx.data_ptr(), k=kx.data_ptr(), v=kx.data_ptr(),
                     qs=H * d * lq, kvs=H * d * 40, ldq=lq, ldkv=40, sc=work.data_ptr(), dsc=dsc.data_ptr(), lse=lse2.data_ptr(),
                     delta=dl.data_ptr(), dq=dq_.data_ptr(), dk=dk_.data_ptr(), dv=dv_.data_ptr(), dqs=H * d * lq, dkvs=H * d * 40,
                     E=1, H=H, d=d, nq=lq, nk=lk, Tp=64, p=0.0, seed=0, st=_stream())
        _mutations(lib.csn_cross_attn_bwd_f32, names, valid, [
            ("dctx", None, ARG), ("ctx", None, ARG), ("q", None, ARG), ("sc", None, ARG), ("dsc", None, ARG), ("lse", None, ARG),
            ("delta", None, ARG), ("dq", None, ARG), ("dk", None, ARG), ("dv", None, ARG), ("d", 24, DIM), ("nq", 26, ALIGN),
            ("nk", 0, ARG), ("nk", 44, ARG), ("nq", 28, ARG), ("p", 1.0, ARG), ("E", 0, ARG), ("H", -1, ARG),
            ("dq", dq_.data_ptr() + 4, PTR), ("cs", H * d * lq + 2, STRIDE)])

        # out-projection + LayerNorm forward
        E, Cm, D, NP = 1, 64, 64, 40
        att = torch.randn(E, D, NP, device=dev); wfc = torch.randn(Cm, D, device=dev); xr = torch.randn(E, Cm, NP, device=dev)
        xh = torch.empty(E, Cm, NP, device=dev); rs = torch.empty(E, NP, device=dev)
        names = ["ctx", "cs", "wfc", "xres", "xs", "ri", "xhat", "xhs", "rstd", "E", "C", "D", "ld", "np", "eps", "p", "seed", "sum", "ws",
                 "wsn", "st"]
        valid = dict(ctx=att.data_ptr(), cs=D * NP, wfc=wfc.data_ptr(), xres=xr.data_ptr(), xs=Cm * NP, ri=None, xhat=xh.data_ptr(),
                     xhs=Cm * NP, rstd=rs.data_ptr(), E=E, C=Cm, D=D, ld=NP, np=NP, eps=1e-6, p=0.0, seed=0, sum=None, ws=None, wsn=0,
                     st=_stream())
        _mutations(lib.csn_outproj_ln_fwd_f32, names, valid, [
            ("ctx", None, ARG), ("wfc", None, ARG), ("xres", None, ARG), ("xhat", None, ARG), ("rstd", None, ARG), ("C", 48, DIM),
            ("C", 512, DIM), ("ld", 42, ALIGN), ("np", 38, ALIGN), ("ld", 36, ARG), ("D", 62, ALIGN), ("cs", D * NP + 2, STRIDE), ("xs", Cm * NP + 1, STRIDE),
            ("E", 0, ARG), ("np", 0, ARG), ("p", 1.0, ARG), ("ctx", att.data_ptr() + 4, PTR), ("np", 44, ARG)])

        # ... and backward (xhat / rstd of the valid forward call above)
        dxh = torch.randn(E, Cm, NP, device=dev); dz = torch.empty(E, Cm, NP, device=dev); dzr = torch.empty(E, Cm, NP, device=dev)
        dat = torch.empty(E, D, NP, device=dev); dwf = torch.empty(Cm, D, device=dev); wt = wfc.t().contiguous()
        wsn2 = lib.csn_wgrad_workspace_floats(Cm, D, E, NP)
        ws2 = torch.empty(max(wsn2, 4), device=dev)
        names = ["dxhat", "xhat", "rstd", "es", "ctx", "cs", "wt", "dz", "dzr", "dctx", "dw", "ws", "wsn", "E", "C", "D", "ld", "np", "acc", "p",
                 "seed", "split", "ps", "rows", "nd", "scale", "grp", "st"]
        valid = dict(dxhat=dxh.data_ptr(), xhat=xh.data_ptr(), rstd=rs.data_ptr(), es=Cm * NP, ctx=att.data_ptr(), cs=D * NP,
                     wt=wt.data_ptr(), dz=dz.data_ptr(), dzr=dzr.data_ptr(), dctx=dat.data_ptr(), dw=dwf.data_ptr(), ws=ws2.data_ptr(),
                     wsn=wsn2, E=E, C=Cm, D=D, ld=NP, np=NP, acc=0, p=0.0, seed=0, split=0, ps=0, rows=None, nd=E, scale=None, grp=0,
                     st=_stream())
        cases = [("xhat", None, ARG), ("rstd", None, ARG), ("ctx", None, ARG), ("wt", None, ARG), ("dz", None, ARG), ("dctx", None, ARG),
                 ("dw", None, ARG), ("ws", None, ARG), ("dxhat", None, ARG), ("E", 0, ARG), ("np", 0, ARG), ("np", 44, ARG), ("ld", 42, ALIGN),
                 ("D", 62, ALIGN), ("es", Cm * NP + 2, STRIDE), ("p", 1.0, ARG), ("nd", 2, ARG), ("grp", -1, ARG), ("split", 1, ARG),
                 ("dz", dz.data_ptr() + 4, PTR), ("C", 48, None)]
        if wsn2 > 0:
            cases.append(("wsn", wsn2 - 1, WS))
        _mutations(lib.csn_outproj_ln_bwd_f32, names, valid, cases)

        # projection weight gradient: workspace too small
        dout = torch.randn(S, R, N, device=dev); dw = torch.empty(R, C, device=dev)
        ws_n = lib.csn_wgrad_workspace_floats(R, C, S, N)
        ws = torch.empty(max(ws_n, 4), device=dev)
        names = ["dout", "ds", "ldd", "x", "xs", "ldx", "dw", "rows", "ch", "ns", "np", "scale", "acc", "ws", "wsn", "st"]
        valid = dict(dout=dout.data_ptr(), ds=R * N, ldd=N, x=x.data_ptr(), xs=C * N, ldx=N, dw=dw.data_ptr(), rows=R, ch=C, ns=S,
                     np=N, scale=1.0, acc=0, ws=ws.data_ptr(), wsn=ws_n, st=_stream())
        cases = [("dout", None, ARG), ("x", None, ARG), ("dw", None, ARG), ("ldd", N + 1, ALIGN), ("np", N - 1, ALIGN), ("ldx", N - 4, ARG),
                 ("ds", R * N + 1, STRIDE), ("ns", 0, ARG), ("rows", 0, ARG)]
        if ws_n > 0:
            cases += [("wsn", ws_n - 1, WS), ("ws", None, None)]
        _mutations(lib.csn_project_wgrad_f32, names, valid, cases)

        # retrieval measure, pooled sums, mix
        f1 = torch.randn(2, 50, 64, device=dev); f2 = torch.randn(3, 30, 64, device=dev); r = torch.empty(2, 3, device=dev)
        need = 2 * 50 + 3 * 30 + 2 * 3 * 50
        wsr = torch.empty(need, device=dev)
        names = ["f1", "f2", "out", "s1", "n1", "s2", "n2", "ch", "ws", "wsn", "st"]
        valid = dict(f1=f1.data_ptr(), f2=f2.data_ptr(), out=r.data_ptr(), s1=2, n1=50, s2=3, n2=30, ch=64, ws=wsr.data_ptr(), wsn=need,
                     st=_stream())
        _mutations(lib.csn_retrieval_measure_f32, names, valid, [
            ("f1", None, ARG), ("f2", None, ARG), ("out", None, ARG), ("ws", None, None), ("wsn", need - 1, WS), ("s1", 0, ARG),
            ("n2", 0, ARG), ("ch", 0, None), ("ch", 66, None)])
        xs_ = torch.randn(6, N, device=dev); so = torch.empty(6, device=dev)
        names = ["x", "out", "rows", "np", "ld", "st"]
        valid = dict(x=xs_.data_ptr(), out=so.data_ptr(), rows=6, np=N, ld=N, st=_stream())
        _mutations(lib.csn_rowsum_f32, names, valid, [("x", None, ARG), ("out", None, ARG), ("rows", 0, ARG), ("np", 0, ARG),
                                                      ("ld", N - 4, ARG), ("ld", N + 2, ALIGN)])
        B, K1 = 2, 3
        xh3 = torch.randn(B * K1, C, N, device=dev); comp = torch.rand(B, K1, device=dev)
        gam = torch.randn(C, device=dev); bet = torch.randn(C, device=dev); feats = torch.empty(B, C, N, device=dev)
        names = ["xhat", "comp", "g", "b", "feats", "B", "k1", "ch", "np", "self", "st"]
        valid = dict(xhat=xh3.data_ptr(), comp=comp.data_ptr(), g=gam.data_ptr(), b=bet.data_ptr(), feats=feats.data_ptr(), B=B, k1=K1,
                     ch=C, np=N, self=None, st=_stream())
        _mutations(lib.csn_mix_fwd_f32, names, valid, [("xhat", None, ARG), ("comp", None, ARG), ("feats", None, ARG), ("k1", 9, ARG),
                                                       ("k1", 0, ARG), ("np", 38, ALIGN), ("B", 0, ARG)])
        dfe = torch.randn(B, C, N, device=dev); dxh3 = torch.empty_like(xh3)
        rdot = torch.empty(B, K1, C, device=dev); rsum = torch.empty(B, C, device=dev)
        names = ["dfeats", "xhat", "comp", "g", "dxhat", "rowdot", "rowsum", "B", "k1", "ch", "np", "self", "dself", "st"]
        valid = dict(dfeats=dfe.data_ptr(), xhat=xh3.data_ptr(), comp=comp.data_ptr(), g=gam.data_ptr(), dxhat=dxh3.data_ptr(),
                     rowdot=rdot.data_ptr(), rowsum=rsum.data_ptr(), B=B, k1=K1, ch=C, np=N, self=None, dself=None, st=_stream())
        _mutations(lib.csn_mix_bwd_f32, names, valid, [("dfeats", None, ARG), ("xhat", None, ARG), ("comp", None, ARG), ("g", None, ARG),
                                                       ("rowdot", None, ARG), ("rowsum", None, ARG), ("k1", 9, ARG), ("np", 38, ALIGN),
                                                       ("B", 0, ARG), ("dself", dxh3.data_ptr(), ARG)])
        torch.cuda.synchronize()
    finally:
        lib.csn_set_math_mode(1)


def test_abi_argument_validation_table_round4_entries(L):
    """The entry points added in round 4 — csn_project_qkv_f32, csn_masked_ce_fwd_f32 / _bwd_f32 — refuse what they cannot take
    with the documented status, like the rest of the table above."""
    lib = L.lib()
    ARG, ALIGN, PTR, STRIDE, DIM, WS = -1, -2, -3, -4, -5, -6
    dev = "cuda"
    L.check(lib.csn_set_math_mode(1))
    S, C, D, N, T, nb = 2, 256, 256, 72, 36, 2
    ldp = nb * 1024
    x = torch.randn(S, C, N, device=dev); w = torch.randn(3 * D, C, device=dev)
    q = torch.empty(S, D, N, device=dev); kv = torch.empty(S, 2 * D, ldp, device=dev, dtype=torch.bfloat16)
    names = ["x", "xs", "ldx", "w", "D", "C", "q", "qs", "ldq", "kv", "kvs", "ldkv", "S", "np", "temp", "T", "st"]
    valid = dict(x=x.data_ptr(), xs=C * N, ldx=N, w=w.data_ptr(), D=D, C=C, q=q.data_ptr(), qs=D * N, ldq=N, kv=kv.data_ptr(), kvs=2 * D * ldp,
                 ldkv=ldp, S=S, np=N, temp=16.0, T=T, st=_stream())
    _mutations(lib.csn_project_qkv_f32, names, valid, [
        ("x", None, ARG), ("w", None, ARG), ("q", None, ARG), ("kv", None, ARG), ("D", 0, ARG), ("S", 0, ARG), ("np", 0, ARG), ("T", 0, ARG),
        ("T", 516, ARG), ("T", 34, ARG), ("ldkv", ldp - 8, ARG), ("ldkv", 1024, ARG), ("ldx", N - 4, ARG), ("ldq", N - 4, ARG), ("np", N - 2, ALIGN),
        ("ldx", N + 2, ALIGN), ("x", x.data_ptr() + 4, PTR), ("q", q.data_ptr() + 8, PTR), ("xs", C * N + 2, STRIDE), ("kvs", 2 * D * ldp + 4, STRIDE)])
    L.check(lib.csn_set_math_mode(0))
    try:
        assert lib.csn_project_qkv_f32(*[valid[n] for n in names]) == ARG           # tile planes: the 16-bit modes only
    finally:
        L.check(lib.csn_set_math_mode(1))

    n_cls, NP = 39, 200
    z = torch.randn(S, n_cls + 1, NP, device=dev); lab = torch.randint(0, n_cls, (S, NP), device=dev)
    lse = torch.empty(S, NP, device=dev); stats = torch.empty(3, device=dev)
    wsb = lib.csn_masked_ce_workspace_bytes(S, NP)
    ws = torch.empty(wsb // 8 + 1, device=dev, dtype=torch.float64)
    names = ["z", "zs", "ld", "lab", "ls", "S", "ncls", "np", "mask", "lse", "ws", "wsb", "stats", "st"]
    valid = dict(z=z.data_ptr(), zs=(n_cls + 1) * NP, ld=NP, lab=lab.data_ptr(), ls=NP, S=S, ncls=n_cls, np=NP, mask=0, lse=lse.data_ptr(),
                 ws=ws.data_ptr(), wsb=wsb, stats=stats.data_ptr(), st=_stream())
    _mutations(lib.csn_masked_ce_fwd_f32, names, valid, [
        ("z", None, ARG), ("lab", None, ARG), ("lse", None, ARG), ("ws", None, ARG), ("stats", None, ARG), ("S", 0, ARG), ("ncls", 0, ARG),
        ("np", 0, ARG), ("ld", NP - 4, ARG), ("wsb", wsb - 8, WS), ("ws", ws.data_ptr() + 4, PTR), ("S", 70000, DIM)])
    g = torch.ones(1, device=dev); dz = torch.empty(S, n_cls, NP, device=dev)
    names = ["z", "zs", "ld", "lab", "ls", "S", "ncls", "np", "mask", "lse", "stats", "g", "dz", "dzs", "dld", "st"]
    valid = dict(z=z.data_ptr(), zs=(n_cls + 1) * NP, ld=NP, lab=lab.data_ptr(), ls=NP, S=S, ncls=n_cls, np=NP, mask=0, lse=lse.data_ptr(),
                 stats=stats.data_ptr(), g=g.data_ptr(), dz=dz.data_ptr(), dzs=n_cls * NP, dld=NP, st=_stream())
    _mutations(lib.csn_masked_ce_bwd_f32, names, valid, [
        ("z", None, ARG), ("lab", None, ARG), ("lse", None, ARG), ("stats", None, ARG), ("g", None, ARG), ("dz", None, ARG), ("S", 0, ARG),
        ("np", 0, ARG), ("dld", NP - 4, ARG), ("ncls", 70000, DIM)])
    # (round 5: the backward takes any point count, pitch and alignment — its one-point-per-thread form — like the forward and like
    #  the reference's loss; tests/test_gpu_loss.py holds those geometries to the oracle)
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,K1,C,ref_layout", [(1, 3, 256, True), (2, 4, 256, True), (32, 4, 256, True), (5, 5, 96, True),
                                               (3, 3, 128, False), (4, 8, 64, True)])
def test_compat_head(L, math_mode, B, K1, C, ref_layout):
    """csn_compat_fwd_f32 / _bwd_f32 against the torch restatement of csa_models.py:222-230 (two nn.Linear, F.normalize, dot,
    softmax; the reference's neighbour-major key bookkeeping for B > 1) in float64: comp and all five gradients.  The head is
    plain fp32 arithmetic in every math mode."""
    import torch.nn.functional as F
    from csn_amd import functional as CF
    rng = np.random.default_rng(77 + B + K1)
    pooled = _rand(rng, B, K1, C)
    wq, wk = _rand(rng, C, C) / math.sqrt(C), _rand(rng, C, C) / math.sqrt(C)
    bq, bk = _rand(rng, C) * 0.1, _rand(rng, C) * 0.1
    dcomp = _rand(rng, B, K1)

    def ref(pooled, wq, bq, wk, bk):
        u_q = F.normalize(F.linear(pooled[:, 0], wq, bq), dim=-1)
        keys = pooled.transpose(0, 1).reshape(K1 * B, C).view(B, K1, C) if ref_layout else pooled
        u_k = F.normalize(F.linear(keys, wk, bk), dim=-1)
        return F.softmax(torch.einsum("bc,bkc->bk", u_q, u_k), dim=-1)

    a64 = [t.double().clone().requires_grad_(True) for t in (pooled, wq, bq, wk, bk)]
    c64 = ref(*a64)
    c64.backward(dcomp.double())
    a32 = [t.cuda().clone().requires_grad_(True) for t in (pooled, wq, bq, wk, bk)]
    comp = CF.compat_head(*a32, reference_layout=ref_layout)
    comp.backward(dcomp.cuda())
    torch.cuda.synchronize()
    assert (comp.detach().cpu().double() - c64.detach()).abs().max().item() < 2e-6
    assert abs(comp.sum(dim=1).detach().cpu() - 1).max().item() < 1e-6
    for got, want, name in zip(a32, a64, ("pooled", "wq", "bq", "wk", "bk")):
        assert _maxerr(got.grad, want.grad) < 2e-5, name
    again = [t.detach().clone().requires_grad_(True) for t in a32]                 # sums over shapes in a fixed order: bitwise repeatable
    CF.compat_head(*again, reference_layout=ref_layout).backward(dcomp.cuda())
    assert all(torch.equal(x.grad, y.grad) for x, y in zip(a32, again))


@pytest.mark.parametrize("S,C,N,R,div_rows", [(9, 260, 236, 200, 100), (2, 100, 1004, 460, 300), (3, 256, 10000, 256, 256)])
@pytest.mark.parametrize("mode", [1, 2])
def test_sixteen_wave_gemm_equals_the_eight_wave_one(L, mode, S, C, N, R, div_rows):
    """The plain 256 x 256 products run on the 16-wave kernel in the bf16x3 mode (and in the one-plane modes under the
    development switch): the same bits as the 8-wave kernel — same slabs, same products in the same order — on ragged tiles."""
    from csn_amd import functional as CF
    lib = L.lib()
    L.check(lib.csn_set_math_mode(mode))
    rng = np.random.default_rng(3)
    x, w = _rand(rng, S, C, N).cuda(), (_rand(rng, R, C) / math.sqrt(C)).cuda()
    try:
        outs = []
        for wide in (0, 2):
            lib.csn_dev_set(L.DEV_WX, 0)
            lib.csn_dev_set(L.DEV_WIDE_GEMM, wide)
            outs.append(CF.project(x, w, div_rows=div_rows, temperature=16.0).clone())
        assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])
    finally:
        lib.csn_dev_set(L.DEV_WIDE_GEMM, 1)
        lib.csn_dev_set(L.DEV_WX, L.DEV_WX_DEFAULT)
        lib.csn_set_math_mode(1)


def test_sixteen_wave_forms_leave_the_step_bit_for_bit(L):
    """The whole module step in bf16x3 with the 16-wave GEMM forms (plain, grouped tile-plane dV / dK, weight gradients) against
    the 8-wave kernel: logits and all 11 gradients bit for bit (same slabs, same products, same order; train mode, same masks)."""
    from csn_amd.csa_models import get_model
    from oracle import csa_oracle as orc
    lib = L.lib()
    L.check(lib.csn_set_math_mode(1))
    rng = np.random.default_rng(61)
    B, K, n_cls, C, N = 2, 2, 7, 256, 1500
    torch.manual_seed(2)
    model = get_model("csa", n_cls, 1, K, block=500, n_blocks=3).cuda().train()
    nbf = torch.from_numpy(rng.standard_normal((B, K + 1, C, N, 1)).astype(np.float32)).cuda()
    x = nbf[:, 0].contiguous()
    lab = torch.from_numpy(rng.integers(0, n_cls, size=(B, N))).cuda()
    outs = []
    try:
        lib.csn_dev_set(L.DEV_WX, 0)          # (the streaming kernel sums in another order: compared against the oracle, not bitwise)
        for forms in (0, 7):
            lib.csn_dev_set(L.DEV_WIDE_FORMS, forms)
            for prm in model.parameters():
                prm.grad = None
            torch.manual_seed(4)
            logits = model(x, "train", nbf)
            orc.masked_ce_loss(logits, lab).backward()
            outs.append((logits.detach().clone(), [p.grad.clone() for p in model.parameters() if p.grad is not None]))
    finally:
        lib.csn_dev_set(L.DEV_WIDE_FORMS, 7)
        lib.csn_dev_set(L.DEV_WX, L.DEV_WX_DEFAULT)
    assert torch.equal(outs[0][0], outs[1][0]) and len(outs[0][1]) == 11
    assert all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))
