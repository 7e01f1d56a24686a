"""The weight-stationary streaming kernel (csn_amd/csrc/wx_stream.hip) behind csn_project_f32 and the dCtx product of
csn_outproj_ln_bwd_f32 in the bf16x3 mode, K = 256 (MID-FC/csa_models.py:103-105, 115): against float64 on the host, against the
tiled GEMM kernels it replaces (switch CSN_DEV_WX), on ragged point counts, strided slots, several row sets, and with the
output carved to end exactly where its allocation ends."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from csn_amd import _lib
    _lib.build()
    lib = _lib.lib()
    _lib.check(lib.csn_set_math_mode(1))
    assert lib.csn_dev_get(_lib.DEV_WX) == _lib.DEV_WX_DEFAULT
    return _lib


def _rand(rng, *shape):
    return torch.from_numpy(rng.standard_normal(size=shape).astype(np.float32))


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _rel(got, ref):
    return ((got.double().cpu() - ref).abs().max() / ref.abs().max()).item()


@pytest.mark.parametrize("S,N,ld,R,div_rows,temp", [
    (2, 1000, 1000, 256, 256, 16.0),        # 31 chunks + a short one of 8 points
    (3, 36, 40, 512, 0, 1.0),               # two chunks per slot, row pitch above the point count
    (1, 10000, 10000, 768, 256, 16.0),      # the reference's geometry, Q | K | V rows in one call (three row sets)
    (5, 500, 512, 256, 256, math.sqrt(96)), # a temperature that is not a power of two: true division
    (300, 64, 64, 256, 0, 1.0),             # more slots than streams
    (1, 4, 4, 2048, 1024, 16.0),            # eight row sets (H = 8), one chunk of 4 points
])
def test_projection_against_float64_and_the_tiled_kernel(L, S, N, ld, R, div_rows, temp):
    lib = L.lib()
    rng = np.random.default_rng(5)
    C = 256
    x = torch.zeros((S, C, ld))
    x[:, :, :N] = _rand(rng, S, C, N)
    w = _rand(rng, R, C) / math.sqrt(C)
    xd, wd = x.cuda(), w.cuda()
    ref = torch.einsum("rc,scn->srn", w.double(), x[:, :, :N].double())
    ref[:, :div_rows] /= temp
    outs = []
    for wx in (1, 0):
        lib.csn_dev_set(L.DEV_WX, wx)
        try:
            out = torch.full((S, R, ld), float("nan"), device="cuda")
            L.check(lib.csn_project_f32(xd.data_ptr(), C * ld, ld, wd.data_ptr(), R, C, out.data_ptr(), R * ld, ld, S, N, div_rows,
                                        temp, 0, 0, _stream()))
            outs.append(out.cpu())
        finally:
            lib.csn_dev_set(L.DEV_WX, L.DEV_WX_DEFAULT)
    for out in outs:
        assert _rel(out[:, :, :N], ref) < 2e-5
        assert torch.isnan(out[:, :, N:]).all()                       # nothing beyond the points of a row is written
    # the same products in the same order (k steps of 16, small terms first): bit for bit the tiled kernel's result
    assert torch.equal(outs[0][:, :, :N], outs[1][:, :, :N])


@pytest.mark.parametrize("S,T,nb,N,R", [(2, 500, 2, 1000, 512), (1, 100, 3, 300, 256), (3, 36, 2, 72, 512), (1, 500, 3, 1300, 512)])
def test_tile_planes_against_float64_and_the_tiled_kernel(L, S, T, nb, N, R):
    """K / V leave as bf16 tile planes: per row and block 16 tiles of [hi 32 | lo 32]; padding keys of the block's last tile are
    zeros, tiles beyond it stay untouched; a row that ends inside the last block (N < nb * T)."""
    lib = L.lib()
    rng = np.random.default_rng(6)
    C = 256
    x, w = _rand(rng, S, C, N), _rand(rng, R, C) / math.sqrt(C)
    xd, wd = x.cuda(), w.cuda()
    ldp = nb * 1024
    ref = torch.einsum("rc,scn->srn", w.double(), x.double())
    outs = []
    for wx in (1, 0):
        lib.csn_dev_set(L.DEV_WX, wx)
        try:
            kv = torch.full((S, R, ldp), float("nan"), device="cuda", dtype=torch.bfloat16)
            L.check(lib.csn_project_f32(xd.data_ptr(), C * N, N, wd.data_ptr(), R, C, kv.data_ptr(), R * ldp, ldp, S, N, 0, 1.0, 2, T,
                                        _stream()))
            outs.append(kv.view(S, R, nb, 16, 2, 32).float().cpu())
        finally:
            lib.csn_dev_set(L.DEV_WX, L.DEV_WX_DEFAULT)
    assert torch.equal(torch.isnan(outs[0]), torch.isnan(outs[1]))     # the same elements are written by both kernels
    assert torch.equal(torch.nan_to_num(outs[0]), torch.nan_to_num(outs[1]))   # ... with the same bits
    for t in outs:
        got = (t[..., 0, :] + t[..., 1, :]).reshape(S, R, nb, 512)
        for b in range(nb):
            n_b = min(T, N - b * T)                                    # points of this block (the last one may be short)
            assert _rel(got[:, :, b, :n_b], ref[:, :, b * T:b * T + n_b]) < 3e-5
            last = (n_b + 31) // 32 * 32
            assert (got[:, :, b, n_b:last] == 0).all()
            assert torch.isnan(got[:, :, b, last:]).all()


def test_strided_slots_and_an_output_that_ends_its_allocation(L):
    """Slot ranges (first, step, count) as the sharded plans pass them, and the output map carved so that its last row ends the
    tensor's storage, followed by a poisoned guard: not one byte of the guard may change (buffer windows, not luck)."""
    lib = L.lib()
    rng = np.random.default_rng(7)
    C, N, R, S, step = 256, 100, 256, 7, 2
    x, w = _rand(rng, S, C, N), _rand(rng, R, C) / math.sqrt(C)
    xd, wd = x.cuda(), w.cuda()
    count = (S + step - 1) // step
    pool = torch.full((count * R * N + 4096,), float("nan"), device="cuda")     # output map + guard in ONE allocation
    out, guard = pool[:count * R * N].view(count, R, N), pool[count * R * N:]
    L.check(lib.csn_project_f32(xd.data_ptr(), step * C * N, N, wd.data_ptr(), R, C, out.data_ptr(), R * N, N, count, N, 0, 1.0, 0, 0,
                                _stream()))
    ref = torch.einsum("rc,scn->srn", w.double(), x[::step].double())
    assert _rel(out, ref) < 2e-5
    assert torch.isnan(guard).all()


def test_dctx_product_inside_the_layer_norm_backward(L):
    """csn_outproj_ln_bwd_f32: dCtx = W_fc^T dZ on the streaming kernel equals the tiled kernel's to fp32 rounding (dZ and the
    weight gradient are the same launches in both)."""
    lib = L.lib()
    rng = np.random.default_rng(8)
    E, C, D, NP = 3, 256, 256, 520
    dxhat, xhat = _rand(rng, E, C, NP).cuda(), _rand(rng, E, C, NP).cuda()
    rstd = (torch.rand(E, NP) + 0.5).cuda()
    ctx = _rand(rng, E, D, NP).cuda()
    wfc_t = (_rand(rng, D, C) / 16).cuda()
    res = []
    for wx in (1, 0):
        lib.csn_dev_set(L.DEV_WX, wx)
        try:
            dz, dctx, dw = torch.empty((E, C, NP), device="cuda"), torch.full((E, D, NP), float("nan"), device="cuda"), torch.empty((C, D), device="cuda")
            ws_n = lib.csn_wgrad_workspace_floats(C, D, E, NP)
            ws = torch.empty((ws_n,), device="cuda")
            L.check(lib.csn_outproj_ln_bwd_f32(dxhat.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), C * NP, ctx.data_ptr(), D * NP,
                                               wfc_t.data_ptr(), dz.data_ptr(), None, dctx.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws_n,
                                               E, C, D, NP, NP, 0, 0.0, 0, 0, 0, None, E, None, 1, _stream()))
            res.append((dz.cpu(), dctx.cpu(), dw.cpu()))
        finally:
            lib.csn_dev_set(L.DEV_WX, L.DEV_WX_DEFAULT)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][2], res[1][2])
    ref = torch.einsum("dc,ecn->edn", wfc_t.double().cpu(), res[0][0].double())
    assert _rel(res[0][1], ref) < 2e-5 and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("E,S,NP,ld,p,indexed", [
    (3, 2, 520, 520, 0.0, True),            # 17 chunks per evaluation (the last of 8 points), evaluations sharing residual slots
    (2, 2, 1000, 1000, 0.1, False),         # fc dropout: the mask of the tiled kernel, bit for bit, or xhat is nowhere near
    (300, 7, 64, 64, 0.0, True),            # more evaluations than streams: runs that cross several evaluations
    (1, 1, 10000, 10000, 0.25, False),      # the reference's point count: one evaluation shared by all the streams
    (5, 5, 36, 40, 0.0, False),             # row pitch above the point count
])
def test_out_projection_layer_norm_on_the_stream(L, E, S, NP, ld, p, indexed):
    """csn_outproj_ln_fwd_f32 (csa_models.py:115-118) on the streaming kernel: xhat, rstd and the pooled sums against float64 and
    against the 256 x 256-tile kernel it replaces (CSN_DEV_WX bit 2 switches the route off); nothing written outside the maps."""
    lib = L.lib()
    rng = np.random.default_rng(17)
    C = D = 256
    seed = 0x1234_5678_9abc_def0
    ctx = torch.zeros((E, D, ld))
    ctx[:, :, :NP] = _rand(rng, E, D, NP)
    x = torch.zeros((S, C, ld))
    x[:, :, :NP] = _rand(rng, S, C, NP)
    wfc = _rand(rng, C, D) / math.sqrt(D)
    rid = torch.from_numpy(rng.integers(0, S, size=E).astype(np.int32)) if indexed else None
    cd, xd, wd = ctx.cuda(), x.cuda(), wfc.cuda()
    ridd = rid.cuda() if indexed else None
    ws_n = lib.csn_outproj_ln_workspace_floats(E, C, D, NP)
    assert ws_n >= E * ((NP + 255) // 256) * C
    res = []
    for wx, fused in ((1, True), (5, True), (1, False)):
        lib.csn_dev_set(L.DEV_WX, wx)
        try:
            pool = torch.full((E * C * ld + 4096,), float("nan"), device="cuda")
            xhat, guard = pool[:E * C * ld].view(E, C, ld), pool[E * C * ld:]
            rstd = torch.full((E * NP + 1024,), float("nan"), device="cuda")
            sums = torch.full((E, C), float("nan"), device="cuda")
            ws = torch.full((ws_n,), float("nan"), device="cuda") if fused else None
            L.check(lib.csn_outproj_ln_fwd_f32(cd.data_ptr(), D * ld, wd.data_ptr(), xd.data_ptr(), C * ld,
                                               ridd.data_ptr() if indexed else None, xhat.data_ptr(), C * ld, rstd.data_ptr(), E, C, D,
                                               ld, NP, 1e-6, p, seed, sums.data_ptr(), ws.data_ptr() if fused else None,
                                               ws_n if fused else 0, _stream()))
            torch.cuda.synchronize()
            assert torch.isnan(guard).all() and torch.isnan(rstd[E * NP:]).all()
            if ld > NP:
                assert torch.isnan(xhat[:, :, NP:]).all()
            res.append((xhat[:, :, :NP].cpu(), rstd[:E * NP].view(E, NP).cpu(), sums.cpu()))
        finally:
            lib.csn_dev_set(L.DEV_WX, L.DEV_WX_DEFAULT)
    (xh, rs, sm), (xh_t, rs_t, sm_t), (xh_n, rs_n, sm_n) = res
    assert torch.equal(xh, xh_n) and torch.equal(rs, rs_n)         # with or without the fused sums: the same maps
    assert (xh - xh_t).abs().max() < 2e-5 and ((rs - rs_t).abs() / rs_t).max() < 2e-5
    tol = 2e-5 * math.sqrt(NP) * 4
    assert (sm - sm_t).abs().max() < tol and (sm - sm_n).abs().max() < tol
    assert (sm.double() - xh.double().sum(dim=2)).abs().max() < tol
    if p == 0.0:
        z = torch.einsum("cd,edn->ecn", wfc.double(), ctx[:, :, :NP].double()) + (x[rid.long()] if indexed else x)[:, :, :NP].double()
        ref = (z - z.mean(dim=1, keepdim=True)) / torch.sqrt(z.var(dim=1, unbiased=False, keepdim=True) + 1e-6)
        assert _rel(xh, ref) < 5e-6
        assert ((rs.double() - 1 / torch.sqrt(z.var(dim=1, unbiased=False) + 1e-6)).abs() * torch.sqrt(z.var(dim=1, unbiased=False))).max() < 1e-5


@pytest.mark.parametrize("S,N,T,nb,temp", [(3, 1000, 500, 2, 16.0), (2, 1300, 500, 3, math.sqrt(200.0)), (5, 72, 36, 2, 16.0),
                                           (130, 500, 500, 1, 16.0)])
def test_q_k_v_in_one_pass_equal_the_two_projections_bit_for_bit(L, S, N, T, nb, temp):
    """csn_project_qkv_f32 (csa_models.py:103-105 on one input): one launch, three row sets walking the same chunks of x, against
    csn_project_f32 twice — every bit of Qs and of the K | V tile planes, padding included, nothing written beyond."""
    lib = L.lib()
    rng = np.random.default_rng(31)
    C = D = 256
    x = _rand(rng, S, C, N).cuda()
    w = (_rand(rng, 3 * D, C) / 16).cuda()
    ldp = nb * 1024
    res = []
    for one in (True, False):
        qpool = torch.full((S * D * N + 1024,), float("nan"), device="cuda")
        kvpool = torch.full((S * 2 * D * ldp + 2048,), 7.0, device="cuda", dtype=torch.bfloat16)
        q, kv = qpool[:S * D * N].view(S, D, N), kvpool[:S * 2 * D * ldp].view(S, 2 * D, ldp)
        if one:
            L.check(lib.csn_project_qkv_f32(x.data_ptr(), C * N, N, w.data_ptr(), D, C, q.data_ptr(), D * N, N, kv.data_ptr(), 2 * D * ldp, ldp,
                                            S, N, temp, T, _stream()))
        else:
            L.check(lib.csn_project_f32(x.data_ptr(), C * N, N, w.data_ptr(), D, C, q.data_ptr(), D * N, N, S, N, D, temp, 0, 0, _stream()))
            L.check(lib.csn_project_f32(x.data_ptr(), C * N, N, w[D:].data_ptr(), 2 * D, C, kv.data_ptr(), 2 * D * ldp, ldp, S, N, 0, 1.0, 2, T,
                                        _stream()))
        torch.cuda.synchronize()
        assert torch.isnan(qpool[S * D * N:]).all() and (kvpool[S * 2 * D * ldp:] == 7.0).all()
        res.append((q.cpu(), kv.cpu().view(torch.int16)))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    ref = torch.einsum("rc,scn->srn", w[:D].double().cpu(), x.double().cpu()) / temp
    assert _rel(res[0][0], ref) < 2e-5


def test_layer_norm_backward_in_groups_of_evaluations_is_the_same_call(L):
    """csn_outproj_ln_bwd_f32 alternates LayerNorm backward and dCtx over groups of evaluations (CSN_DEV_LNB_GROUP): every output
    of a call in groups of 2 (5 evaluations: 2 + 2 + 1) equals the one-launch call bit for bit — the dropout masks, the scale
    rows and the shared gradient maps are indexed by the evaluation's own number, not by its place in the group."""
    lib = L.lib()
    rng = np.random.default_rng(12)
    E, C, D, NP, grp = 5, 256, 256, 260, 2
    dfeats = _rand(rng, (E + grp - 1) // grp, C, NP).cuda()
    scale, rows = _rand(rng, E, C).cuda(), (_rand(rng, E, C) / NP).cuda()
    xhat, rstd, ctx = _rand(rng, E, C, NP).cuda(), (torch.rand(E, NP) + 0.5).cuda(), _rand(rng, E, D, NP).cuda()
    wfc_t = (_rand(rng, D, C) / 16).cuda()
    res = []
    for wx, G in ((wx, G) for wx in (1, 9) for G in (0, 2)):         # the two launches, then the fused kernel (wx_lnb.hip)
        prev, prev_wx = lib.csn_dev_set(L.DEV_LNB_GROUP, G), lib.csn_dev_set(L.DEV_WX, wx)
        try:
            dz, dzr = torch.full((E, C, NP), float("nan"), device="cuda"), torch.full((E, C, NP), float("nan"), device="cuda")
            dctx, dw = torch.full((E, D, NP), float("nan"), device="cuda"), torch.empty((C, D), device="cuda")
            ws_n = lib.csn_wgrad_workspace_floats(C, D, E, NP)
            ws = torch.empty((ws_n,), device="cuda")
            L.check(lib.csn_outproj_ln_bwd_f32(dfeats.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), C * NP, ctx.data_ptr(), D * NP,
                                               wfc_t.data_ptr(), dz.data_ptr(), dzr.data_ptr(), dctx.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws_n,
                                               E, C, D, NP, NP, 0, 0.2, 0x5eed, 0, 0, rows.data_ptr(), E - 1, scale.data_ptr(), grp, _stream()))
            torch.cuda.synchronize()
            res.append([t.cpu() for t in (dz, dzr, dctx, dw)])
        finally:
            lib.csn_dev_set(L.DEV_LNB_GROUP, prev)
            lib.csn_dev_set(L.DEV_WX, prev_wx)
    for one, grouped in ((res[0], res[1]), (res[2], res[3])):
        for a, b in zip(one, grouped):
            assert not torch.isnan(a).any() and torch.equal(a, b)
    assert (res[0][0] == 0).float().mean() > 0.1                      # the fc dropout mask is live in dz, and absent from dz_res
    assert (res[0][1] == 0).float().mean() < 0.01


@pytest.mark.parametrize("E,NP,p,with_res,grp", [(5, 260, 0.2, True, 2), (3, 1000, 0.0, False, 1), (40, 96, 0.1, True, 8), (2, 10000, 0.1, False, 2)])
def test_layer_norm_backward_fused_into_the_dctx_stream(L, E, NP, p, with_res, grp):
    """csn_outproj_ln_bwd_f32 with the LayerNorm backward computed on the chunks' way into the dCtx product (wx_lnb.hip, CSN_DEV_WX
    bit 3) against the two launches it replaces: dz (the same dropout mask: the same zeros), dz_res, dCtx and the weight gradient to
    fp32 rounding (the row sums are formed in another order), against float64, twice the same bits, nothing outside the maps."""
    lib = L.lib()
    rng = np.random.default_rng(21)
    C = D = 256
    n_src = (E + grp - 1) // grp
    dfeats = _rand(rng, n_src, C, NP).cuda()
    scale, rows = _rand(rng, E, C).cuda(), (_rand(rng, E, C) / 8).cuda()
    xh = _rand(rng, E, C, NP)
    xh = (xh - xh.mean(dim=1, keepdim=True)) / xh.std(dim=1, unbiased=False, keepdim=True)       # what a LayerNorm leaves
    xhat, rstd, ctx = xh.cuda(), (torch.rand(E, NP) + 0.5).cuda(), _rand(rng, E, D, NP).cuda()
    wfc_t = (_rand(rng, D, C) / 16).cuda()
    n_dense = E - 1
    res = []
    for wx in (9, 9, 1):
        prev = lib.csn_dev_set(L.DEV_WX, wx)
        try:
            pool = torch.full((3 * E * C * NP + 4096,), float("nan"), device="cuda")
            dz, dzr, dctx = (pool[i * E * C * NP:(i + 1) * E * C * NP].view(E, C, NP) for i in range(3))
            dw = torch.empty((C, D), device="cuda")
            ws_n = lib.csn_wgrad_workspace_floats(C, D, E, NP)
            ws = torch.empty((ws_n,), device="cuda")
            L.check(lib.csn_outproj_ln_bwd_f32(dfeats.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), C * NP, ctx.data_ptr(), D * NP,
                                               wfc_t.data_ptr(), dz.data_ptr(), dzr.data_ptr() if with_res else None, dctx.data_ptr(), dw.data_ptr(),
                                               ws.data_ptr(), ws_n, E, C, D, NP, NP, 0, p, 0xabcdef, 0, 0, rows.data_ptr(), n_dense, scale.data_ptr(),
                                               grp, _stream()))
            torch.cuda.synchronize()
            assert torch.isnan(pool[3 * E * C * NP:]).all()
            if not with_res:
                assert torch.isnan(dzr).all()
            res.append([t.cpu().clone() for t in (dz, dzr, dctx, dw)])
        finally:
            lib.csn_dev_set(L.DEV_WX, prev)
    a, a2, b = res
    for t, u in zip(a, a2):
        assert torch.equal(torch.nan_to_num(t, nan=7.0), torch.nan_to_num(u, nan=7.0))
    assert not torch.isnan(a[0]).any() and not torch.isnan(a[2]).any()
    assert torch.equal(a[0] == 0, b[0] == 0)                                    # the mask
    for i in ((0, 1, 2, 3) if with_res else (0, 2, 3)):
        assert (a[i] - b[i]).abs().max() <= 2e-5 * b[i].abs().max(), i
    # float64: the LayerNorm backward itself (dz_res is dz without the mask)
    dx = torch.zeros((E, C, NP), dtype=torch.float64)
    src = torch.arange(E) // grp
    dx[:n_dense] = dfeats.cpu().double()[src[:n_dense]] * scale.cpu().double()[:n_dense, :, None]
    dx += rows.cpu().double()[:, :, None]
    x64 = xh.double()
    ref = rstd.cpu().double()[:, None, :] * (dx - dx.mean(dim=1, keepdim=True) - x64 * (dx * x64).mean(dim=1, keepdim=True))
    got = a[1] if with_res else a[0]
    keep = torch.ones_like(ref, dtype=torch.bool) if with_res else (a[0] != 0)
    sc = 1.0 if with_res else 1.0 / (1.0 - p)
    assert ((got.double() - ref * sc).abs()[keep]).max() <= 2e-5 * ref.abs().max() * sc
