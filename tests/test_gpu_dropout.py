"""Train-mode dropout of the CSA attention (csa_models.py:141 on the probabilities, :115 on the fc output) in the HIP
path.  The mask is counter-based, so it can be restated exactly on the host (tests/dropout_ref.py): forward AND
backward are checked element-wise against a float64 reference that uses that very mask; the module-level check
against the reference's own torch.nn.Dropout stream is statistical (different RNG)."""
import math

import numpy as np
import pytest
import torch

from oracle import csa_oracle as orc
from tests import dropout_ref as dr

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def exact_fp32_mode():
    """The element-wise checks against a float64 reference with the very same mask use fp32-rounding tolerances: they run in
    the exact-fp32 math mode (the masks themselves do not depend on the mode)."""
    from csn_amd import _lib
    _lib.build()
    _lib.check(_lib.lib().csn_set_math_mode(0))
    yield
    _lib.lib().csn_set_math_mode(1)


def _rand(rng, *shape):
    return torch.from_numpy(rng.standard_normal(size=shape).astype(np.float32))


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _rel(got, ref):
    ref = ref.double()
    return ((got.detach().cpu().double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("E,H,d,T,nb,p", [(2, 2, 64, 100, 2, 0.1), (1, 1, 256, 500, 1, 0.1), (2, 1, 32, 36, 3, 0.5)])
def test_attention_dropout_forward_backward_with_host_mask(E, H, d, T, nb, p):
    from csn_amd import _lib as L
    L.build()
    rng = np.random.default_rng(21)
    D, N, Tp, seed = H * d, T * nb, (T + 31) // 32 * 32, 0x1234_5678_9abc_def1
    q = _rand(rng, E, D, N) / math.sqrt(math.sqrt(d))
    k = _rand(rng, E, D, N) / math.sqrt(math.sqrt(d))
    v, dctx = _rand(rng, E, D, N), _rand(rng, E, D, N)
    qd, kd, vd, dd = q.cuda(), k.cuda(), v.cuda(), dctx.cuda()
    ctx = torch.empty((E, D, N), device="cuda")
    lse = torch.empty((E, H, N), device="cuda")
    scores = torch.empty((E, H, nb, T, Tp), device="cuda")
    L.check(L.lib().csn_block_attn_fwd_f32(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), D * N, D * N, None, None, N,
                                           ctx.data_ptr(), D * N, scores.data_ptr(), lse.data_ptr(), E, H, d, T, nb, Tp, 8.0,
                                           p, seed, 0, 0, _stream()))
    mask = torch.from_numpy(dr.attention_mask(E, H, nb, T, Tp, seed, p)).double()      # [e][h][blk][key][query]
    assert abs(mask.mean().item() - (1 - p)) < 4 * math.sqrt(p * (1 - p) / mask.numel())

    q64, k64, v64 = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    qb = q64.view(E, H, d, nb, T).permute(0, 1, 3, 4, 2)
    kb = k64.view(E, H, d, nb, T).permute(0, 1, 3, 4, 2)
    vb = v64.view(E, H, d, nb, T).permute(0, 1, 3, 4, 2)
    pr = torch.softmax(qb @ kb.transpose(-1, -2), dim=-1)                               # [e][h][blk][query][key]
    pdrop = pr * mask.transpose(-1, -2) / (1 - p)
    ref = (pdrop @ vb).permute(0, 1, 4, 2, 3).reshape(E, D, N)
    assert _rel(ctx, ref) < 5e-6
    ref.backward(dctx.double())

    dscores = torch.empty_like(scores)
    delta = torch.empty((E, H, N), device="cuda")
    dq, dk, dv = (torch.empty((E, D, N), device="cuda") for _ in range(3))
    L.check(L.lib().csn_block_attn_bwd_dq_f32(dd.data_ptr(), ctx.data_ptr(), D * N, kd.data_ptr(), vd.data_ptr(), D * N, None,
                                              N, scores.data_ptr(), dscores.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                              dq.data_ptr(), D * N, None, 0, None, E, H, d, T, nb, Tp, p, seed, 0, 0, 0, 0, 0, None, 0, _stream()))
    L.check(L.lib().csn_block_attn_bwd_dkv_f32(dd.data_ptr(), D * N, qd.data_ptr(), D * N, None, N, scores.data_ptr(),
                                               dscores.data_ptr(), dk.data_ptr(), dv.data_ptr(), D * N, None, None, 0, None, E, H,
                                               d, T, nb, Tp, 0, 0, 0, 0, 0, None, 0, _stream()))
    torch.cuda.synchronize()
    assert _rel(dq, q64.grad) < 2e-5 and _rel(dk, k64.grad) < 2e-5 and _rel(dv, v64.grad) < 2e-5
    # the scores buffer now holds the dropped probabilities
    assert (scores[..., :T].cpu().double() - pdrop.detach()).abs().max().item() < 2e-6

    # another seed gives another mask; the same seed reproduces the output bit-for-bit
    ctx2, ctx3 = torch.empty_like(ctx), torch.empty_like(ctx)
    for out, sd in ((ctx2, seed + 1), (ctx3, seed)):
        L.check(L.lib().csn_block_attn_fwd_f32(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), D * N, D * N, None, None, N,
                                               out.data_ptr(), D * N, None, lse.data_ptr(), E, H, d, T, nb, Tp, 8.0, p, sd, 0, 0,
                                               _stream()))
    torch.cuda.synchronize()
    assert torch.equal(ctx3, ctx) and not torch.equal(ctx2, ctx)


@pytest.mark.parametrize("E,C,D,NP,p", [(2, 256, 256, 1000, 0.1), (1, 96, 192, 500, 0.3)])
def test_fc_dropout_forward_backward_with_host_mask(E, C, D, NP, p):
    from csn_amd import _lib as L
    L.build()
    rng = np.random.default_rng(22)
    seed = 0x0fed_cba9_8765_4321
    att, wfc, x, dxhat = _rand(rng, E, D, NP), _rand(rng, C, D) / math.sqrt(D), _rand(rng, E, C, NP), _rand(rng, E, C, NP)
    attd, wd, xd, dxd = att.cuda(), wfc.cuda(), x.cuda(), dxhat.cuda()
    xhat = torch.empty((E, C, NP), device="cuda")
    rstd = torch.empty((E, NP), device="cuda")
    L.check(L.lib().csn_outproj_ln_fwd_f32(attd.data_ptr(), D * NP, wd.data_ptr(), xd.data_ptr(), C * NP, None, xhat.data_ptr(),
                                           C * NP, rstd.data_ptr(), E, C, D, NP, NP, 1e-6, p, seed, None, None, 0, _stream()))
    mask = torch.from_numpy(dr.fc_mask(E, C, NP, seed, p)).double()
    a64, w64, x64 = att.double().requires_grad_(True), wfc.double().requires_grad_(True), x.double().requires_grad_(True)
    z = torch.einsum("cd,edn->ecn", w64, a64) * mask / (1 - p) + x64
    ref = (z - z.mean(dim=1, keepdim=True)) / torch.sqrt(z.var(dim=1, unbiased=False, keepdim=True) + 1e-6)
    assert _rel(xhat, ref) < 5e-6
    ref.backward(dxhat.double())
    dz, dz_res = torch.empty((E, C, NP), device="cuda"), torch.empty((E, C, NP), device="cuda")
    datt, dw = torch.empty((E, D, NP), device="cuda"), torch.empty((C, D), device="cuda")
    ws_n = L.lib().csn_wgrad_workspace_floats(C, D, E, NP)
    ws = torch.empty((ws_n,), device="cuda")
    wt = wd.t().contiguous()
    L.check(L.lib().csn_outproj_ln_bwd_f32(dxd.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), C * NP, attd.data_ptr(), D * NP,
                                           wt.data_ptr(), dz.data_ptr(), dz_res.data_ptr(), datt.data_ptr(), dw.data_ptr(),
                                           ws.data_ptr(), ws_n, E, C, D, NP, NP, 0, p, seed, 0, 0, None, E, None, 1, _stream()))
    torch.cuda.synchronize()
    assert _rel(datt, a64.grad) < 2e-5 and _rel(dw, w64.grad) < 2e-5 and _rel(dz_res, x64.grad) < 2e-5


def test_train_mode_module_matches_reference_statistics():
    """model.train(): dropout live like csa_training.py:192.  The reference draws its masks from torch's generator, we
    from a counter hash, so parity is statistical: moments of the outputs over many seeds."""
    from csn_amd.csa_models import MultiHeadAttention
    rng = np.random.default_rng(23)
    C, H, N, T, n_seeds = 64, 2, 400, 100, 48
    p = orc.make_params(rng, H, d_model=C, d_k=32, d_v=32, csa=False)
    m = MultiHeadAttention(H, C, 32, 32, block=T, n_blocks=4).cuda().train()
    m.load_state_dict({k[len("attention."):]: v for k, v in p.items() if k.startswith("attention.")})
    x = orc.synth_points(rng, (1, C, N, 1))
    xd = x.cuda()
    ours, theirs = [], []
    torch.manual_seed(1)
    with torch.no_grad():
        for _ in range(n_seeds):
            ours.append(m(xd, xd, xd, "train")[0].cpu())
        for _ in range(n_seeds):
            theirs.append(_oracle_train(x, p, H, T, 4))
    ours, theirs = torch.stack(ours), torch.stack(theirs)
    assert not torch.equal(ours[0], ours[1])                                      # masks change from call to call
    m_eval = m.eval()
    with torch.no_grad():
        y_eval = m_eval(xd, xd, xd, "test")[0].cpu()
    # (1) seed-mean per element agrees within the standard error, (2) per-element spread agrees, (3) both differ from eval
    se = (ours.std(0) ** 2 / n_seeds + theirs.std(0) ** 2 / n_seeds).sqrt()
    zscore = (ours.mean(0) - theirs.mean(0)) / se.clamp_min(1e-6)
    assert zscore.abs().mean().item() < 1.2 and (zscore.abs() > 4.5).float().mean().item() < 1e-3
    assert abs(ours.std(0).mean().item() / theirs.std(0).mean().item() - 1.0) < 0.05
    assert (ours.mean(0) - y_eval).abs().mean().item() < 3 * ours.std(0).mean().item()


def _oracle_train(x, p, H, T, nb, p_attn=0.1, p_fc=0.1):
    """closed-form oracle with torch dropout at the two reference sites (csa_models.py:141, :115)."""
    import torch.nn.functional as F
    xs = x.squeeze(-1).permute(0, 2, 1)
    B, N, C = xs.shape
    d = p["attention.w_qs.weight"].shape[0] // H
    xb = xs.reshape(B, nb, T, C)
    heads = lambda t: t.reshape(B, nb, T, H, d).transpose(-3, -2)
    q, k, v = (heads(F.linear(xb, p[f"attention.{n}.weight"])) for n in ("w_qs", "w_ks", "w_vs"))
    pr = F.dropout(torch.softmax((q / d ** 0.5) @ k.transpose(-1, -2), dim=-1), p_attn, training=True)
    ctx = (pr @ v).transpose(-3, -2).reshape(B, nb, T, H * d)
    z = F.dropout(F.linear(ctx, p["attention.fc.weight"]), p_fc, training=True) + xb
    return F.layer_norm(z, (C,), p["attention.norm.weight"], p["attention.norm.bias"], 1e-6).reshape(B, N, C)


def test_csa_model_trains_in_train_mode():
    """One csa_training.py-style step with model.train(): runs, finite, gradients for the 11 trained tensors; the pooled
    self evaluation and the mixed self evaluation draw different masks (csa_models.py:210 vs :232)."""
    from csn_amd.csa_models import get_model
    rng = np.random.default_rng(24)
    B, K, n_cls = 2, 2, 6
    p = orc.make_params(rng, 1, n_cls=n_cls, csa=True)
    model = get_model("csa", n_cls, 1, K)
    model.load_state_dict(p, strict=False)
    model = model.cuda().train()
    x = orc.synth_points(rng, (B, 256, 10000, 1))
    nb = orc.synth_points(rng, (B, K + 1, 256, 10000, 1))
    nb[:, 0] = x
    lab = orc.synth_labels(rng, B, 10000, n_cls).cuda()
    torch.manual_seed(3)
    l1 = orc.masked_ce_loss(model(x.cuda(), "train", nb), lab)
    l1.backward()
    grads = {n: q.grad for n, q in model.named_parameters() if q.grad is not None}
    assert len(grads) == 11 and all(torch.isfinite(g).all() for g in grads.values())
    torch.manual_seed(3)
    l2 = orc.masked_ce_loss(model(x.cuda(), "train", nb), lab)
    l3 = orc.masked_ce_loss(model(x.cuda(), "train", nb), lab)
    assert l1.item() == l2.item() and l3.item() != l2.item()                     # torch.manual_seed reproduces the masks
    with torch.no_grad():
        le = orc.masked_ce_loss(model.eval()(x.cuda(), "test", nb), lab)
    assert abs(l1.item() - le.item()) < 0.2 * le.item()
    assert model.attention.plan("csa_train", B, K + 1, x.cuda().device).E == B * (2 * K + 2)   # 2K+2 evaluations per shape
