"""GPU parity of the MinkowskiNet-variant attention (csn_amd/minkowski_attention.py): cross-length,
point-major, gradients to queries / keys / values and to all weights — against the float64 oracle restatement
(oracle.mha_pointmajor), in both math modes."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import csn_amd
    csn_amd.build()
    from csn_amd import _lib
    return _lib


@pytest.fixture(autouse=True, params=[0, 1], ids=["fp32", "bf16x3"])
def math_mode(request, L):
    L.check(L.lib().csn_set_math_mode(request.param))
    yield request.param
    L.lib().csn_set_math_mode(1)


def _params(rng, H, C, d):
    from oracle import csa_oracle as orc
    return orc.make_params(rng, H, d_model=C, d_k=d, d_v=d)


def _module(p, H, C, d):
    from csn_amd.minkowski_attention import MultiHeadAttention
    m = MultiHeadAttention(H, C, d, d)
    m.load_state_dict({k[len("attention."):]: v for k, v in p.items() if k.startswith("attention.")}, strict=False)
    return m.cuda()


@pytest.mark.parametrize("b,lq,lk,H,C", [(2, 100, 64, 4, 256), (1, 37, 53, 4, 256), (1, 1000, 1301, 4, 256), (2, 48, 48, 1, 128),
                                         (1, 130, 7, 2, 64), (1, 90, 61, 5, 256), (2, 75, 40, 3, 128)])
def test_cross_length_mha_forward_backward(L, math_mode, b, lq, lk, H, C):
    """(the last two cases: d = d_model // n_head = 51 and 42 — hrnet.py:343 builds whatever that division gives — run at the
    64-wide kernel instance with zero-padded weights)"""
    from oracle import csa_oracle as orc
    rng = np.random.default_rng(17)
    d = C // H
    p = _params(rng, H, C, d)
    m = _module(p, H, C, d).eval()
    q = torch.from_numpy(rng.standard_normal((b, lq, C)).astype(np.float32))
    k = torch.from_numpy(rng.standard_normal((b, lk, C)).astype(np.float32))
    v = torch.from_numpy(rng.standard_normal((b, lk, C)).astype(np.float32))
    qd, kd, vd = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out, attn = m(qd, kd, vd)
    assert out.shape == (b, lq, C) and attn.shape == (b, H, lq, lk)
    g = torch.from_numpy(rng.standard_normal((b, lq, C)).astype(np.float32))
    out.backward(g.cuda())

    p64 = {n: t.double().requires_grad_(True) for n, t in p.items() if n.startswith("attention.")}
    q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
    ref, rattn = orc.mha_pointmajor(q64, k64, v64, p64, H, d, d)
    ref.backward(g.double())
    def rel(got, want):
        return ((got.detach().cpu().double() - want).abs().max() / want.abs().max().clamp_min(1e-30)).item()

    e_out = (out.detach().cpu().double() - ref.detach()).abs().max().item()
    e_attn = (attn.cpu().double() - rattn.detach()).abs().max().item()
    e_in = {n: rel(a.grad, b.grad) for n, a, b in (("dq", qd, q64), ("dk", kd, k64), ("dv", vd, v64))}
    e_w = {name: rel(prm.grad, p64["attention." + name].grad) for name, prm in m.named_parameters()}
    print(f"[minkowski] mode {math_mode} b={b} lq={lq} lk={lk} H={H} C={C}: out {e_out:.1e} attn {e_attn:.1e} "
          + " ".join(f"{n} {e:.1e}" for n, e in {**e_in, **e_w}.items()))
    # the 1e-4 contract in BOTH math modes (measured in bf16x3: outputs <= 5e-6, every gradient <= 1.3e-5)
    assert e_out < 1e-4 and e_attn < 1e-4
    assert max(e_in.values()) < 1e-4 and max(e_w.values()) < 1e-4, (e_in, e_w)


def test_self_call_shares_input(L, math_mode):
    """MHA(x, x, x) — hrnet.py:464 get_SSA: the same tensor serves as queries, keys and values, its gradient is the sum."""
    from oracle import csa_oracle as orc
    rng = np.random.default_rng(18)
    H, C, n = 4, 256, 203
    d = C // H
    p = _params(rng, H, C, d)
    m = _module(p, H, C, d).eval()
    x = torch.from_numpy(rng.standard_normal((1, n, C)).astype(np.float32))
    xd = x.cuda().requires_grad_(True)
    out, _ = m(xd, xd, xd)
    out.square().sum().backward()
    x64 = x.double().requires_grad_(True)
    p64 = {k_: t.double() for k_, t in p.items()}
    ref, _ = orc.mha_pointmajor(x64, x64, x64, p64, H, d, d)
    ref.square().sum().backward()
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() < 1e-4
    assert ((xd.grad.cpu().double() - x64.grad).abs().max() / x64.grad.abs().max()).item() < 1e-4


def test_standalone_sdpa_cross_lengths(L, math_mode):
    from csn_amd.minkowski_attention import ScaledDotProductAttention
    rng = np.random.default_rng(19)
    B, H, lq, lk, d = 2, 3, 45, 70, 64
    q, k, v = (torch.from_numpy(rng.standard_normal(s).astype(np.float32)) for s in ((B, H, lq, d), (B, H, lk, d), (B, H, lk, d)))
    att = ScaledDotProductAttention(math.sqrt(d)).eval()
    qd, kd, vd = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out, prob = att(qd, kd, vd)
    out.sum().backward()
    q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
    rp = torch.softmax((q64 / math.sqrt(d)) @ k64.transpose(2, 3), dim=-1)
    ro = rp @ v64
    ro.sum().backward()
    assert (out.detach().cpu().double() - ro.detach()).abs().max().item() < 1e-4
    assert (prob.cpu().double() - rp.detach()).abs().max().item() < 2e-5
    for got, want in ((qd.grad, q64.grad), (kd.grad, k64.grad), (vd.grad, v64.grad)):
        assert ((got.cpu().double() - want).abs().max() / want.abs().max()).item() < 1e-4


def test_train_mode_dropout_statistics(L, math_mode):
    """Both dropouts live: outputs stay finite, differ between calls, and average towards the eval-mode output."""
    rng = np.random.default_rng(20)
    H, C, lq, lk = 4, 256, 64, 80
    p = _params(rng, H, C, C // H)
    m = _module(p, H, C, C // H)
    q = torch.from_numpy(rng.standard_normal((1, lq, C)).astype(np.float32)).cuda()
    k = torch.from_numpy(rng.standard_normal((1, lk, C)).astype(np.float32)).cuda()
    m.eval()
    ref, _ = m(q, k, k)
    m.train()
    torch.manual_seed(0)
    outs = torch.stack([m(q, k, k)[0] for _ in range(24)])
    assert torch.isfinite(outs).all() and (outs[0] - outs[1]).abs().max().item() > 1e-3
    assert (outs.mean(0) - ref).abs().mean().item() < 0.15


def test_ragged_batch_in_one_launch_chain(L, math_mode):
    """forward_varlen: five shape pairs, every query and key count different (down to 1 key, up to several query tiles),
    against one ``forward`` call per pair (the way hrnet.py:378-410 runs the layer) — outputs and the gradients to every
    input and weight — and against the float64 oracle."""
    from oracle import csa_oracle as orc
    rng = np.random.default_rng(29)
    H, C = 4, 256
    d = C // H
    p = _params(rng, H, C, d)
    lens = [(7, 301), (45, 70), (1301, 37), (512, 500), (100, 1)]
    qs = [torch.from_numpy(rng.standard_normal((n, C)).astype(np.float32)) for n, _ in lens]
    ks = [torch.from_numpy(rng.standard_normal((m, C)).astype(np.float32)) for _, m in lens]
    vs = [torch.from_numpy(rng.standard_normal((m, C)).astype(np.float32)) for _, m in lens]
    gs = [torch.from_numpy(rng.standard_normal((n, C)).astype(np.float32)) for n, _ in lens]

    def run(varlen):
        m = _module(p, H, C, d).eval()
        qd, kd, vd = ([t.cuda().requires_grad_(True) for t in ts] for ts in (qs, ks, vs))
        if varlen:
            outs = m.forward_varlen(qd, kd, vd)
        else:
            outs = [m(q[None], k[None], v[None])[0][0] for q, k, v in zip(qd, kd, vd)]
        sum((o * g.cuda()).sum() for o, g in zip(outs, gs)).backward()
        return ([o.detach().cpu() for o in outs], [[t.grad.cpu() for t in ts] for ts in (qd, kd, vd)],
                {n: q.grad.cpu() for n, q in m.named_parameters()})

    (o1, g1, w1), (o0, g0, w0) = run(True), run(False)
    for i, (n, mm) in enumerate(lens):
        assert o1[i].shape == (n, C)
        assert (o1[i] - o0[i]).abs().max().item() < 2e-6, i                     # same kernels, same arithmetic per pair
        gt = 2e-5 if math_mode == 0 else 1e-4      # bf16x3: the padded batch takes other GEMM tilings (each ~1e-5 from exact)
        for a, b_ in zip((g1[0][i], g1[1][i], g1[2][i]), (g0[0][i], g0[1][i], g0[2][i])):
            # (pair 4 has ONE key: its softmax is the constant 1, so dq and dk are pure rounding noise there, ~1e-5 of the
            #  other gradients' size: differences are measured against at least 1e-1)
            assert (a - b_).abs().max().item() <= gt * max(b_.abs().max().item(), 1e-1), i
    for n_ in w0:
        assert (w1[n_] - w0[n_]).abs().max().item() <= gt * w0[n_].abs().max().item(), n_
    # and the oracle, pair by pair
    p64 = {n_: t.double() for n_, t in p.items() if n_.startswith("attention.")}
    for i in range(len(lens)):
        ref, _ = orc.mha_pointmajor(qs[i][None].double(), ks[i][None].double(), vs[i][None].double(), p64, H, d, d)
        assert (o1[i].double() - ref[0]).abs().max().item() < 1e-4, i
    # train mode: finite, different from eval
    m = _module(p, H, C, d).train()
    torch.manual_seed(1)
    outs = m.forward_varlen([t.cuda() for t in qs], [t.cuda() for t in ks], [t.cuda() for t in vs])
    assert all(torch.isfinite(o).all() for o in outs) and (outs[3].cpu() - o1[3]).abs().max().item() > 1e-3


def test_unequal_head_widths(L, math_mode):
    """d_k = 40, d_v = 72 (attention.py:12 takes both; hrnet.py always passes equal ones): outputs and every gradient against
    the float64 oracle."""
    from oracle import csa_oracle as orc
    from csn_amd.minkowski_attention import MultiHeadAttention
    rng = np.random.default_rng(31)
    b, lq, lk, H, C, dk, dv = 2, 83, 131, 3, 96, 40, 72
    p = orc.make_params(rng, H, d_model=C, d_k=dk, d_v=dv)
    m = MultiHeadAttention(H, C, dk, dv)
    m.load_state_dict({k[len("attention."):]: v for k, v in p.items() if k.startswith("attention.")}, strict=False)
    m = m.cuda().eval()
    q, k, v = (torch.from_numpy(rng.standard_normal(s).astype(np.float32)) for s in ((b, lq, C), (b, lk, C), (b, lk, C)))
    g = torch.from_numpy(rng.standard_normal((b, lq, C)).astype(np.float32))
    qd, kd, vd = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out, attn = m(qd, kd, vd)
    assert attn.shape == (b, H, lq, lk)
    out.backward(g.cuda())
    p64 = {n: t.double().requires_grad_(True) for n, t in p.items() if n.startswith("attention.")}
    q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
    ref, rattn = orc.mha_pointmajor(q64, k64, v64, p64, H, dk, dv)
    ref.backward(g.double())
    rel = lambda got, want: ((got.detach().cpu().double() - want).abs().max() / want.abs().max()).item()
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() < 1e-4
    assert (attn.cpu().double() - rattn.detach()).abs().max().item() < 1e-4
    for a, r in ((qd, q64), (kd, k64), (vd, v64)):
        assert rel(a.grad, r.grad) < 1e-4
    for name, prm in m.named_parameters():
        assert prm.grad.shape == prm.shape and rel(prm.grad, p64["attention." + name].grad) < 1e-4, name
