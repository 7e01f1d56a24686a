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
                                         (1, 130, 7, 2, 64)])
def test_cross_length_mha_forward_backward(L, math_mode, b, lq, lk, H, C):
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
    tol = 1e-4 if math_mode == 0 else 2e-4
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() < tol
    assert (attn.cpu().double() - rattn.detach()).abs().max().item() < tol

    def rel(got, want):
        return ((got.detach().cpu().double() - want).abs().max() / want.abs().max().clamp_min(1e-30)).item()

    gtol = 1e-4 if math_mode == 0 else 3e-4
    assert rel(qd.grad, q64.grad) < gtol and rel(kd.grad, k64.grad) < gtol and rel(vd.grad, v64.grad) < gtol
    for name, prm in m.named_parameters():
        assert rel(prm.grad, p64["attention." + name].grad) < gtol, name


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
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() < 2e-4
    assert ((xd.grad.cpu().double() - x64.grad).abs().max() / x64.grad.abs().max()).item() < 3e-4


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
    assert (out.detach().cpu().double() - ro.detach()).abs().max().item() < 2e-4
    assert (prob.cpu().double() - rp.detach()).abs().max().item() < 2e-5
    for got, want in ((qd.grad, q64.grad), (kd.grad, k64.grad), (vd.grad, v64.grad)):
        assert ((got.cpu().double() - want).abs().max() / want.abs().max()).item() < 3e-4


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
