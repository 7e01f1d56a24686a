"""The stand-alone ``ScaledDotProductAttention`` of the drop-in surface (MID-FC/csa_models.py:128-144) on the MI355X:
the reference's own G1 goldens, and the train-mode contract (the DROPPED probabilities are returned, :141-144)."""
import os

import numpy as np
import pytest
import torch

from tests import dropout_ref as dr

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[0, 1], ids=["fp32", "bf16x3"])
def math_mode(request):
    from csn_amd import _lib
    _lib.build()
    _lib.check(_lib.lib().csn_set_math_mode(request.param))
    yield request.param
    _lib.lib().csn_set_math_mode(1)


def test_g1_goldens_through_the_drop_in_class(golden_dir, math_mode):
    """G1 = the reference's ScaledDotProductAttention.eval() on (1,1,500,256), (2,8,500,256), (1,2,64,32): outputs 1e-4,
    probabilities 1e-6 in the exact-fp32 mode and 5e-6 in bf16x3 (a score error of 2e-5 on a probability of 0.06; measured
    1.3e-6)."""
    from oracle import csa_oracle as orc
    from csn_amd.csa_models import ScaledDotProductAttention
    g = np.load(os.path.join(golden_dir, "g1_sdpa.npz"))
    for tag in "abc":
        B, H, T, d, seed = (int(v) for v in g[f"g1{tag}_shape"])
        rng = np.random.default_rng(seed)
        q, k, v = (orc.synth_points(rng, (B, H, T, d)) for _ in range(3))
        m = ScaledDotProductAttention(temperature=d ** 0.5).eval()
        with torch.no_grad():
            o, pr = m(q.cuda(), k.cuda(), v.cuda())
        assert o.shape == (B, H, T, d) and pr.shape == (B, H, T, T)
        assert np.abs(o.cpu().reshape(B * H, T, d)[:, ::31].numpy() - g[f"g1{tag}_out"]).max() < 1e-4
        assert np.abs(pr.cpu().reshape(B * H, T, T)[:, ::31].numpy() - g[f"g1{tag}_prob"]).max() < (5e-6 if math_mode else 1e-6)


@pytest.mark.parametrize("Tq,Tk", [(700, 100), (64, 64), (37, 301)])
def test_train_mode_returns_the_dropped_probabilities(Tq, Tk, math_mode):
    """attn = dropout(softmax(q k^T / t)) and out = attn v (csa_models.py:141-142): the returned probabilities carry the
    kernel's own mask — rebuilt on the host from (seed, position) — and the output is their product with v.  (700, 100):
    more queries than the score pitch (128), where the pair index must still be unique."""
    from csn_amd.csa_models import ScaledDotProductAttention
    rng = np.random.default_rng(7)
    B, H, d, p = 2, 2, 32, 0.1
    q, k, v = (torch.from_numpy(rng.standard_normal(s).astype(np.float32)) for s in ((B, H, Tq, d), (B, H, Tk, d), (B, H, Tk, d)))
    m = ScaledDotProductAttention(temperature=d ** 0.5).train()
    torch.manual_seed(5)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    torch.manual_seed(5)
    with torch.no_grad():
        o, pr = m(q.cuda(), k.cuda(), v.cuda())
    Tp = (Tk + 31) // 32 * 32
    Tq4 = (Tq + 3) // 4 * 4
    mask = dr.attention_mask(B * H, 1, 1, Tk, Tp, seed, p, Tq=Tq4)[:, 0, 0]                 # [e][key][query]
    mask = torch.from_numpy(mask[:, :, :Tq]).permute(0, 2, 1).reshape(B, H, Tq, Tk).double()
    soft = torch.softmax((q.double() / d ** 0.5) @ k.double().transpose(2, 3), dim=-1)
    want = soft * mask / (1.0 - p)
    assert abs(mask.mean().item() - (1.0 - p)) < 0.01
    assert (pr.cpu().double() - want).abs().max().item() < (1e-5 if math_mode else 2e-6)      # bf16x3: measured 3.3e-6
    assert (o.cpu().double() - want @ v.double()).abs().max().item() < 1e-4
    if Tq > Tp:
        # the old pair index (key pair * pitch + query) gave query Tp + a, pair w the mask of query a, pair w + 1
        a = mask[..., :Tq - Tp, 2:]
        b = mask[..., Tp:, :Tk - 2]
        assert (a == b).double().mean().item() < 0.9            # independent masks agree on 0.82 of the positions


def test_unequal_and_odd_head_widths(math_mode):
    """q / k of width 48 and v of width 80 (no kernel instance for either): out, probabilities and the three input gradients
    against float64."""
    import math
    from csn_amd.csa_models import ScaledDotProductAttention
    rng = np.random.default_rng(77)
    B, H, Tq, Tk, dk, dv = 2, 3, 70, 45, 48, 80
    q, k, v = (torch.from_numpy(rng.standard_normal(s).astype(np.float32)) for s in ((B, H, Tq, dk), (B, H, Tk, dk), (B, H, Tk, dv)))
    att = ScaledDotProductAttention(temperature=math.sqrt(dk)).cuda().eval()
    qd, kd, vd = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out, prob = att(qd, kd, vd)
    assert out.shape == (B, H, Tq, dv) and prob.shape == (B, H, Tq, Tk)
    g = torch.from_numpy(rng.standard_normal((B, H, Tq, dv)).astype(np.float32))
    (out * g.cuda()).sum().backward()
    q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
    rp = torch.softmax((q64 / math.sqrt(dk)) @ k64.transpose(2, 3), dim=-1)
    ro = rp @ v64
    (ro * g.double()).sum().backward()
    assert (out.detach().cpu().double() - ro.detach()).abs().max().item() < 1e-4
    assert (prob.cpu().double() - rp.detach()).abs().max().item() < 2e-5
    for got, want in ((qd.grad, q64.grad), (kd.grad, k64.grad), (vd.grad, v64.grad)):
        assert got.shape == want.shape
        assert ((got.cpu().double() - want).abs().max() / want.abs().max()).item() < 1e-4


def test_backward_runs_in_the_mode_of_its_forward():
    """A forward under a THREAD override (``CF.math_mode('fp32')`` while the process default is bf16x3) gets its backward —
    which autograd runs on its own worker thread — in the same arithmetic: the gradients are the bits of a run whose
    process default is fp32, not those of the bf16x3 default."""
    from csn_amd import _lib, functional as CF
    from csn_amd.csa_models import ScaledDotProductAttention
    rng = np.random.default_rng(11)
    B, H, T, d = 1, 2, 96, 32
    q0, k0, v0 = (torch.from_numpy(rng.standard_normal((B, H, T, d)).astype(np.float32)).cuda() for _ in range(3))
    w = torch.from_numpy(rng.standard_normal((B, H, T, d)).astype(np.float32)).cuda()
    m = ScaledDotProductAttention(temperature=d ** 0.5).eval()

    def grads(process_mode, override):
        _lib.check(_lib.lib().csn_set_math_mode(process_mode))
        q, k, v = (t.clone().requires_grad_(True) for t in (q0, k0, v0))
        with CF.math_mode(override):
            o, _ = m(q, k, v)
        (o * w).sum().backward()                       # outside the block: the worker thread has no override of its own
        torch.cuda.synchronize()
        return [t.grad.clone() for t in (q, k, v)]

    exact = grads(0, None)
    scoped = grads(1, "fp32")
    fast = grads(1, None)
    assert all(torch.equal(a, b) for a, b in zip(exact, scoped))
    assert any(not torch.equal(a, b) for a, b in zip(exact, fast))
