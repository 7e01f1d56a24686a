"""The score-recomputing ("flash") data flows of the attention backward against the kept-scores flow and a float64
restatement (ScaledDotProductAttention, MID-FC/csa_models.py:138-144): the forward keeps only lse, the dQ call rebuilds
S = Qs K^T tile by tile (csn_block_attn_bwd_dq_recompute_f32), and — flow FLASH — a key-stationary kernel rebuilds P / dS
for dK / dV (csn_block_attn_bwd_dkv_flash_f32).  Same arithmetic in the same order as the kept-scores kernels, so the flows
agree to fp32 rounding; the float64 bound is the mode's own (1e-4 contract in bf16x3, reported error in bf16)."""
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


@pytest.fixture(autouse=True)
def _restore_mode(L):
    yield
    L.lib().csn_set_math_mode(1)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _rand(rng, *shape):
    return torch.from_numpy(rng.standard_normal(size=shape).astype(np.float32))


def tile_planes(x, T, nb, npl):
    """fp32 (S, R, nb*T) -> the tile planes csn_project_f32(out_split = 2) writes: per row and block 16 tiles of
    [hi 32 | lo 32] bf16 (two planes, math mode 1) or of [32] bf16 (one plane, mode 2); padding keys are zero."""
    S, R, _ = x.shape
    p = torch.zeros((S, R, nb, 512), device=x.device, dtype=torch.float32)
    p[..., :T] = x.view(S, R, nb, T)
    p = p.view(S, R, nb, 16, 32)
    hi = p.bfloat16()
    if npl == 1:
        return hi.reshape(S, R, nb * 512).contiguous()
    lo = (p - hi.float()).bfloat16()
    return torch.stack((hi, lo), dim=4).reshape(S, R, nb * 1024).contiguous()


def _reference(q, k, v, dctx, q_idx, kv_idx, H, d, T, nb, keep=None, keep_scale=1.0):
    """float64 autograd of softmax(q^T k) (-> dropout mask `keep`) v per (evaluation, head, block); returns ctx, dq, dk, dv
    per EVALUATION (E, H*d, N)."""
    E = len(q_idx)
    q64 = q.double()[q_idx].clone().requires_grad_(True)
    k64 = k.double()[kv_idx].clone().requires_grad_(True)
    v64 = v.double()[kv_idx].clone().requires_grad_(True)
    qb = q64.view(E, H, d, nb, T).permute(0, 1, 3, 4, 2)
    kb = k64.view(E, H, d, nb, T).permute(0, 1, 3, 4, 2)
    vb = v64.view(E, H, d, nb, T).permute(0, 1, 3, 4, 2)
    p = torch.softmax(qb @ kb.transpose(-1, -2), dim=-1)
    if keep is not None:
        p = p * keep.double() * keep_scale
    ctx = (p @ vb).permute(0, 1, 4, 2, 3).reshape(E, H * d, nb * T)
    ctx.backward(dctx.double())
    return ctx.detach(), q64.grad, k64.grad, v64.grad


def _err(got, ref):
    ref = ref.double()
    return ((got.detach().cpu().double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


CASES = [
    # mode, S, E, H, d, T, nb, p_drop
    (1, 3, 5, 2, 64, 100, 3, 0.0),
    (1, 2, 3, 1, 128, 500, 2, 0.1),
    (1, 1, 2, 1, 96, 36, 2, 0.1),
    (1, 2, 2, 1, 32, 132, 1, 0.0),
    (2, 2, 3, 1, 256, 500, 2, 0.1),
    (2, 3, 5, 2, 64, 100, 3, 0.0),
    (2, 2, 4, 1, 96, 500, 3, 0.1),
    # the d = 96 one-plane dK / dV kernel keeps TWO 16-key groups per wave (256 keys a work-group): a second key chunk of four
    # keys, a single chunk two thirds empty with two heads, no dropout
    (2, 2, 3, 1, 96, 260, 2, 0.1),
    (2, 1, 2, 2, 96, 132, 1, 0.0),
]


@pytest.mark.parametrize("mode,S,E,H,d,T,nb,p_drop", CASES)
def test_recomputed_scores_equal_the_kept_ones(L, mode, S, E, H, d, T, nb, p_drop):
    lib = L.lib()
    L.check(lib.csn_set_math_mode(mode))
    assert lib.csn_attn_bwd_grouping(d, T) & 4
    rng = np.random.default_rng(100 + d + T)
    D, N, npl = H * d, T * nb, (2 if mode == 1 else 1)
    Tp = (T + 31) // 32 * 32
    q = (_rand(rng, S, D, N) / math.sqrt(math.sqrt(d))).cuda()
    k = _rand(rng, S, D, N) / math.sqrt(math.sqrt(d))
    v = _rand(rng, S, D, N)
    dctx = _rand(rng, E, D, N).cuda()
    q_idx = rng.integers(0, S, size=E).astype(np.int32)
    kv_idx = rng.integers(0, S, size=E).astype(np.int32)
    qi, ki = torch.from_numpy(q_idx).cuda(), torch.from_numpy(kv_idx).cuda()
    kv = tile_planes(torch.cat((k, v), dim=1).cuda(), T, nb, npl)             # (S, 2D, ldp)
    ldp = nb * 512 * npl
    k_ptr, v_ptr, kv_stride = kv.data_ptr(), kv.data_ptr() + 2 * D * ldp, 2 * D * ldp
    seed = 987654321

    def forward(keep):
        ctx = torch.full((E, D, N), float("nan"), device="cuda")
        lse = torch.zeros((E, H, N), device="cuda")
        sc = torch.full((E, H, nb, T, Tp), float("nan"), device="cuda") if keep else None
        L.check(lib.csn_block_attn_fwd_f32(q.data_ptr(), k_ptr, v_ptr, D * N, kv_stride, qi.data_ptr(), ki.data_ptr(), N,
                                           ctx.data_ptr(), D * N, sc.data_ptr() if keep else None, lse.data_ptr(), E, H, d, T,
                                           nb, Tp, 8.0, p_drop, seed, 1, ldp, _stream()), "fwd")
        return ctx, lse, sc

    ctx, lse, scores = forward(True)
    ctx1, lse1, _ = forward(False)
    assert torch.equal(ctx, ctx1) and torch.equal(lse, lse1)                  # the forward is the same with and without the store

    def buffers():
        return (torch.full((E, H, nb, T, Tp), float("nan"), device="cuda"), torch.empty((E, H, N), device="cuda"),
                torch.full((E, D, N), float("nan"), device="cuda"))

    # kept scores: P / dS leave as tile planes (pt = 1)
    ds0, delta0, dq0 = buffers()
    L.check(lib.csn_block_attn_bwd_dq_f32(dctx.data_ptr(), ctx.data_ptr(), D * N, k_ptr, v_ptr, kv_stride, ki.data_ptr(), N,
                                          scores.data_ptr(), ds0.data_ptr(), lse.data_ptr(), delta0.data_ptr(), dq0.data_ptr(),
                                          D * N, None, 0, None, E, H, d, T, nb, Tp, p_drop, seed, 0, 0, 1, ldp, 1, None, 0,
                                          _stream()), "dq kept")
    # recomputed, planes written
    ds1, delta1, dq1 = buffers()
    pr1 = torch.full((E, H, nb, T, Tp), float("nan"), device="cuda")
    L.check(lib.csn_block_attn_bwd_dq_recompute_f32(dctx.data_ptr(), ctx.data_ptr(), D * N, q.data_ptr(), D * N, qi.data_ptr(),
                                                    k_ptr, v_ptr, kv_stride, ki.data_ptr(), N, pr1.data_ptr(), ds1.data_ptr(),
                                                    lse.data_ptr(), delta1.data_ptr(), dq1.data_ptr(), D * N, None, 0, None, E,
                                                    H, d, T, nb, Tp, p_drop, seed, ldp, 0, 1, None, 0, _stream()), "dq recompute")
    # recomputed, nothing written besides delta and dq
    _, delta2, dq2 = buffers()
    L.check(lib.csn_block_attn_bwd_dq_recompute_f32(dctx.data_ptr(), ctx.data_ptr(), D * N, q.data_ptr(), D * N, qi.data_ptr(),
                                                    k_ptr, v_ptr, kv_stride, ki.data_ptr(), N, None, None,
                                                    lse.data_ptr(), delta2.data_ptr(), dq2.data_ptr(), D * N, None, 0, None, E,
                                                    H, d, T, nb, Tp, p_drop, seed, ldp, 0, 0, None, 0, _stream()), "dq recompute, no planes")
    torch.cuda.synchronize()
    assert torch.equal(delta0, delta1) and torch.equal(delta0, delta2)
    # the recomputed S is the forward's product in the forward's order (the same bits: the P planes below are equal); the two
    # kernel instances round the dS formula differently (fused multiply-add contraction), so dS and dQ agree to fp32 rounding
    assert torch.equal(dq1, dq2)                                               # with and without the planes: the same kernel
    rnd = 2e-5 if mode == 1 else 1e-2                                          # (one plane: a flipped bf16 rounding of dS is 2^-9)
    assert _err(dq1, dq0.cpu()) < rnd

    def planes_of(buf, rows):                                                  # 16-bit tile-plane rows -> values
        if mode == 1:
            pl = buf.view(torch.bfloat16).view(E, H, nb, rows, Tp // 32, 2, 32).float()
            return (pl[..., 0, :] + pl[..., 1, :]).reshape(E, H, nb, rows, Tp)
        return buf.view(torch.bfloat16).view(E, H, nb, 2 * T, Tp)[..., :rows, :].float()

    if mode == 1:                                                              # P planes: kept flow in place over the scores
        assert _err(planes_of(pr1, T), planes_of(scores, T).cpu()) < rnd
        assert _err(planes_of(ds1, T), planes_of(ds0, T).cpu()) < rnd
    else:                                                                      # [P rows | dS rows] of 16-bit elements in `dscores`
        a, b = planes_of(ds0, 2 * T), planes_of(ds1, 2 * T)
        assert _err(b[..., :T, :], a[..., :T, :].cpu()) < rnd and _err(b[..., T:, :], a[..., T:, :].cpu()) < rnd
    # ... and the float64 restatement, with the same masks (tests/dropout_ref.py restates the counter-based mask)
    keep = None
    if p_drop > 0:
        from tests.dropout_ref import attention_mask
        keep = torch.from_numpy(attention_mask(E, H, nb, T, Tp, seed, p_drop)).transpose(-1, -2)     # -> [query][key]
    kr, vr = k, v
    if mode == 2:                                                              # one plane: the operands ARE bf16-rounded
        kr, vr = k.bfloat16().float(), v.bfloat16().float()
    _, rq, rk, rv = _reference(q.cpu(), kr, vr, dctx.cpu(), q_idx, kv_idx, H, d, T, nb, keep, 1.0 / (1.0 - p_drop))
    bound = 2e-4 if mode == 1 else 3e-2
    assert _err(dq2, rq) < bound, _err(dq2, rq)

    # dK / dV: the products over the kept flow's P / dS planes against the key-stationary kernel that rebuilds them
    dk0, dv0 = (torch.full((E, D, N), float("nan"), device="cuda") for _ in range(2))
    L.check(lib.csn_block_attn_bwd_dkv_f32(dctx.data_ptr(), D * N, q.data_ptr(), D * N, qi.data_ptr(), N, scores.data_ptr(),
                                           ds0.data_ptr(), dk0.data_ptr(), dv0.data_ptr(), D * N, None, None, 0, None, E, H, d, T,
                                           nb, Tp, 0, 0, 0, 0, 1, None, 0, _stream()), "dkv products")
    torch.cuda.synchronize()
    assert _err(dk0, rk) < bound and _err(dv0, rv) < bound
    flash = bool(lib.csn_attn_bwd_grouping(d, T) & 8)
    assert flash == (d <= 128)
    if flash:
        dk2, dv2 = (torch.full((E, D, N), float("nan"), device="cuda") for _ in range(2))
        L.check(lib.csn_block_attn_bwd_dkv_flash_f32(dctx.data_ptr(), D * N, q.data_ptr(), D * N, qi.data_ptr(), k_ptr, v_ptr,
                                                     kv_stride, ki.data_ptr(), ldp, 0, N, lse.data_ptr(), delta2.data_ptr(),
                                                     dk2.data_ptr(), dv2.data_ptr(), D * N, None, None, 0, None, E, H, d, T, nb,
                                                     Tp, p_drop, seed, None, 0, _stream()), "dkv flash")
        torch.cuda.synchronize()
        assert _err(dk2, rk) < bound and _err(dv2, rv) < bound, (_err(dk2, rk), _err(dv2, rv))
        assert _err(dk2, dk0.cpu()) < rnd * 4 and _err(dv2, dv0.cpu()) < rnd * 4

    # grouped form: the evaluations of a query slot into one set of accumulators
    from csn_amd.functional import EvalPlan
    plan = EvalPlan(q_idx, kv_idx, S, "cuda")
    gq = torch.full((S, D, N), float("nan"), device="cuda")
    L.check(lib.csn_block_attn_bwd_dq_recompute_f32(dctx.data_ptr(), ctx.data_ptr(), D * N, q.data_ptr(), D * N, qi.data_ptr(),
                                                    k_ptr, v_ptr, kv_stride, ki.data_ptr(), N, None, None,
                                                    lse.data_ptr(), delta2.data_ptr(), gq.data_ptr(), D * N, qi.data_ptr(), 0,
                                                    plan.q_group_items.data_ptr(), E, H, d, T, nb, Tp, p_drop, seed, ldp, 0, 0,
                                                    plan.q_group_off.data_ptr(), plan.n_q_groups, _stream()), "dq grouped")
    torch.cuda.synchronize()
    ref = torch.zeros(S, D, N, dtype=torch.float64).index_add_(0, torch.from_numpy(q_idx).long(), rq)
    used = torch.from_numpy(np.unique(q_idx)).long()
    assert _err(gq[used], ref[used]) < bound
    if flash:       # the evaluations of a key/value slot into one set of dK / dV accumulators
        gk, gv = (torch.full((S, D, N), float("nan"), device="cuda") for _ in range(2))
        L.check(lib.csn_block_attn_bwd_dkv_flash_f32(dctx.data_ptr(), D * N, q.data_ptr(), D * N, qi.data_ptr(), k_ptr, v_ptr,
                                                     kv_stride, ki.data_ptr(), ldp, 0, N, lse.data_ptr(), delta2.data_ptr(),
                                                     gk.data_ptr(), gv.data_ptr(), D * N, ki.data_ptr(), ki.data_ptr(), 0,
                                                     plan.kv_group_items.data_ptr(), E, H, d, T, nb, Tp, p_drop, seed,
                                                     plan.kv_group_off.data_ptr(), plan.n_kv_groups, _stream()), "dkv flash grouped")
        torch.cuda.synchronize()
        ki64 = torch.from_numpy(kv_idx).long()
        ref_k = torch.zeros(S, D, N, dtype=torch.float64).index_add_(0, ki64, rk)
        ref_v = torch.zeros(S, D, N, dtype=torch.float64).index_add_(0, ki64, rv)
        used = torch.from_numpy(np.unique(kv_idx)).long()
        assert _err(gk[used], ref_k[used]) < bound and _err(gv[used], ref_v[used]) < bound
        # the same sums colour by colour (accumulate = 1: the epilogue's read-modify-write): every slot written once by the first
        # colour, added to by the others — equal to the grouped call up to the association of the fp32 sums
        ck, cv = (torch.full((S, D, N), float("nan"), device="cuda") for _ in range(2))
        for ci, ids in enumerate(plan.dkv_colors):
            L.check(lib.csn_block_attn_bwd_dkv_flash_f32(dctx.data_ptr(), D * N, q.data_ptr(), D * N, qi.data_ptr(), k_ptr, v_ptr,
                                                         kv_stride, ki.data_ptr(), ldp, 0, N, lse.data_ptr(), delta2.data_ptr(),
                                                         ck.data_ptr(), cv.data_ptr(), D * N, ki.data_ptr(), ki.data_ptr(),
                                                         0 if ci == 0 else 1, ids.data_ptr(), ids.numel(), H, d, T, nb, Tp, p_drop,
                                                         seed, None, 0, _stream()), "dkv flash, one colour")
        torch.cuda.synchronize()
        assert _err(ck[used], gk[used].cpu()) < 1e-5 and _err(cv[used], gv[used].cpu()) < 1e-5


@pytest.mark.parametrize("d,T,nb", [(96, 500, 2), (256, 100, 3)])
def test_fp16_planes_are_converted_while_staged(L, d, T, nb):
    """The backward of an fp16 forward (math mode 3 -> backward in mode 2): K / V planes that hold fp16 bits, flagged kv_f16,
    give exactly the gradients of the same values handed over as bf16 planes — no second projection is needed."""
    lib = L.lib()
    L.check(lib.csn_set_math_mode(2))
    rng = np.random.default_rng(9)
    S = E = 2
    H, D, N = 1, d, T * nb
    Tp = (T + 31) // 32 * 32
    q = (_rand(rng, S, D, N) / math.sqrt(math.sqrt(d))).cuda()
    kvf = torch.cat((_rand(rng, S, D, N) / math.sqrt(math.sqrt(d)), _rand(rng, S, D, N)), dim=1).cuda().half()
    dctx = _rand(rng, E, D, N).cuda()
    idx = torch.arange(E, dtype=torch.int32, device="cuda")
    ldp = nb * 512
    planes16 = tile_planes(kvf.float(), T, nb, 1).view(torch.int16)        # placeholder shape; filled with fp16 bits below
    p = torch.zeros((S, 2 * D, nb, 512), device="cuda", dtype=torch.float16)
    p[..., :T] = kvf.view(S, 2 * D, nb, T)
    planes_f16 = p.reshape(S, 2 * D, ldp).contiguous()
    planes_bf16 = tile_planes(kvf.float(), T, nb, 1)                       # bf16(fp16(x)): what the kernels' conversion yields
    assert planes16.shape == planes_f16.view(torch.int16).shape
    ctx, lse = torch.zeros((E, D, N), device="cuda"), torch.zeros((E, H, N), device="cuda")
    sc = torch.zeros((E, H, nb, T, Tp), device="cuda")
    kp = planes_bf16.data_ptr()
    L.check(lib.csn_block_attn_fwd_f32(q.data_ptr(), kp, kp + 2 * D * ldp, D * N, 2 * D * ldp, idx.data_ptr(), idx.data_ptr(), N,
                                       ctx.data_ptr(), D * N, sc.data_ptr(), lse.data_ptr(), E, H, d, T, nb, Tp, 8.0, 0.1, 77, 1, ldp,
                                       _stream()))
    outs = []
    for planes, f16 in ((planes_bf16, 0), (planes_f16, 1)):
        kp = planes.data_ptr()
        vp = kp + 2 * D * ldp
        res = []
        ds, de, dq = torch.zeros_like(sc), torch.zeros((E, H, N), device="cuda"), torch.zeros((E, D, N), device="cuda")
        s2 = sc.clone()
        L.check(lib.csn_block_attn_bwd_dq_f32(dctx.data_ptr(), ctx.data_ptr(), D * N, kp, vp, 2 * D * ldp, idx.data_ptr(), N,
                                              s2.data_ptr(), ds.data_ptr(), lse.data_ptr(), de.data_ptr(), dq.data_ptr(), D * N, None,
                                              0, None, E, H, d, T, nb, Tp, 0.1, 77, 0, 0, 1 + f16, ldp, 1, None, 0, _stream()), "kept")
        res.append(dq)
        dq2, de2 = torch.zeros((E, D, N), device="cuda"), torch.zeros((E, H, N), device="cuda")
        L.check(lib.csn_block_attn_bwd_dq_recompute_f32(dctx.data_ptr(), ctx.data_ptr(), D * N, q.data_ptr(), D * N, idx.data_ptr(),
                                                        kp, vp, 2 * D * ldp, idx.data_ptr(), N, None, None, lse.data_ptr(),
                                                        de2.data_ptr(), dq2.data_ptr(), D * N, None, 0, None, E, H, d, T, nb, Tp, 0.1,
                                                        77, ldp, f16, 0, None, 0, _stream()), "recompute")
        res.append(dq2)
        if lib.csn_attn_bwd_grouping(d, T) & 8:
            dk, dv = torch.zeros((E, D, N), device="cuda"), torch.zeros((E, D, N), device="cuda")
            L.check(lib.csn_block_attn_bwd_dkv_flash_f32(dctx.data_ptr(), D * N, q.data_ptr(), D * N, idx.data_ptr(), kp, vp,
                                                         2 * D * ldp, idx.data_ptr(), ldp, f16, N, lse.data_ptr(), de2.data_ptr(),
                                                         dk.data_ptr(), dv.data_ptr(), D * N, None, None, 0, None, E, H, d, T, nb, Tp,
                                                         0.1, 77, None, 0, _stream()), "flash")
            res += [dk, dv]
        torch.cuda.synchronize()
        outs.append(res)
    for a, b in zip(*outs):
        assert a.abs().max().item() > 0 and torch.equal(a, b)
    L.check(lib.csn_set_math_mode(1))                                      # two planes: no fp16 forward to pair with
    assert lib.csn_block_attn_bwd_dq_f32(dctx.data_ptr(), ctx.data_ptr(), D * N, kp, vp, 2 * D * ldp, idx.data_ptr(), N, sc.data_ptr(),
                                         ds.data_ptr(), lse.data_ptr(), de.data_ptr(), dq.data_ptr(), D * N, None, 0, None, E, H, d, T,
                                         nb, Tp, 0.1, 77, 0, 0, 2, 2 * ldp, 1, None, 0, _stream()) == -1


def test_recompute_is_refused_where_it_has_no_kernel(L):
    lib = L.lib()
    L.check(lib.csn_set_math_mode(1))
    assert not (lib.csn_attn_bwd_grouping(256, 500) & 4)          # two planes at d = 256: three LDS images do not fit
    z = torch.zeros(64, device="cuda")
    rc = lib.csn_block_attn_bwd_dq_recompute_f32(z.data_ptr(), z.data_ptr(), 0, z.data_ptr(), 0, None, z.data_ptr(),
                                                 z.data_ptr(), 0, None, 36, None, None, z.data_ptr(), z.data_ptr(),
                                                 z.data_ptr(), 0, None, 0, None, 1, 1, 256, 36, 1, 64, 0.0, 0, 1024, 0, 0, None, 0,
                                                 _stream())
    assert rc == -1
    L.check(lib.csn_set_math_mode(0))
    assert lib.csn_attn_bwd_grouping(64, 100) == 0


@pytest.mark.parametrize("mode,geo", [(2, {}), (1, dict(d_model=128, d_k=128, d_v=128, block=100, n_blocks=3)),
                                      (2, dict(d_model=96, d_k=96, d_v=96, block=500, n_blocks=2))],
                         ids=["bf16-d256", "bf16x3-d128", "bf16-d96"])
@pytest.mark.parametrize("train", [False, True], ids=["eval", "train"])
def test_module_flows_agree(L, mode, geo, train):
    """CrossShapeAt forward + loss + backward in every attention-backward data flow the mode has kernels for: logits, loss
    and all 11 gradients equal those of the kept-scores flow (same masks in train mode: same seeds)."""
    from csn_amd import tuning
    from csn_amd.csa_models import get_model
    from oracle import csa_oracle as orc
    L.check(L.lib().csn_set_math_mode(mode))
    rng = np.random.default_rng(77)
    B, K, n_cls = 2, 2, 7
    C = geo.get("d_model", 256)
    N = geo.get("block", 500) * geo.get("n_blocks", 20)
    torch.manual_seed(5)
    model = get_model("csa", n_cls, 1, K, **geo).cuda().train(train)
    off = torch.from_numpy(rng.standard_normal((B, K + 1, C, 1, 1)).astype(np.float32))
    nbf = torch.from_numpy(rng.standard_normal((B, K + 1, C, N, 1)).astype(np.float32)) + 2.0 * off
    x = nbf[:, 0].contiguous()
    lab = torch.from_numpy(rng.integers(0, n_cls, size=(B, N)))
    outs = {}
    flows = [tuning.KEEP_SCORES, tuning.RECOMPUTE_DQ]
    if L.lib().csn_attn_bwd_grouping(C, geo.get("block", 500)) & 8:
        flows.append(tuning.FLASH)
    for flow in flows:
        for prm in model.parameters():
            prm.grad = None
        torch.manual_seed(9)
        with tuning.override(score_flow={1: flow, 2: flow}):
            logits = model(x.cuda(), "train", nbf.cuda())
            loss = orc.masked_ce_loss(logits, lab.cuda())
            loss.backward()
        outs[flow] = (logits.detach(), loss.item(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    l0, s0, g0 = outs[tuning.KEEP_SCORES]
    assert len(g0) == 11
    for flow in flows[1:]:
        l1, s1, g1 = outs[flow]
        assert torch.equal(l0, l1) and s0 == s1                                # the forward is the same kernel, store or no store
        for n in g0:
            scale = g0[n].abs().max().item()
            # same products from other kernel instances (another rounding of the dS formula, another summation order of dK / dV)
            lim = 5e-5 if mode == 1 else 2e-2
            assert (g0[n] - g1[n]).abs().max().item() <= lim * scale, (flow, n)
