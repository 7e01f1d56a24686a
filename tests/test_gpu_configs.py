"""Every BASELINE.json configuration as a workload THROUGH the drop-in module (SURVEY.md §8d numbering):
   config 2  CSA K = 2, 4 shapes x 10000 points x 256 channels           — against the CPU oracle at full size
   config 3  CSA K = 3, 32 x 10000 x 256 (the benched step)               — full size against the CPU oracle (eval-mode arithmetic)
                                                                            and cross-checked between the math modes (train mode)
   config 5  CSA K = 4, 8 x 50000 x 96 in 100 blocks of 500               — against the float64 oracle at 2 x 5000 x 96, at full
                                                                            size against the fp32 oracle, and through
                                                                            size-independent properties
(config 1 is the CPU plumbing case of tests/test_gpu_module.py::test_g2..., config 4 the 8-GPU run of the driver).  Inputs carry
per-shape channel offsets (oracle.conditioned_csa_case) so that all 11 gradients are well-conditioned and held to 1e-4."""
import numpy as np
import pytest
import torch

from oracle import csa_oracle as orc

pytestmark = pytest.mark.gpu


def _set_mode(mode):
    from csn_amd import _lib
    _lib.build()
    _lib.check(_lib.lib().csn_set_math_mode(mode))


@pytest.fixture(autouse=True)
def _restore_mode():
    yield
    _set_mode(1)


def _module(p, n_cls, K, **geo):
    from csn_amd.csa_models import get_model
    m = get_model("csa", n_cls, 1, K, **geo)
    missing, unexpected = m.load_state_dict(p, strict=False)
    assert not unexpected and all(k.startswith("fc_1.") for k in missing)
    return m.cuda()


def _step(model, x, nb, lab):
    for q in model.parameters():
        q.grad = None
    logits = model(x, "test", nb)
    loss = orc.masked_ce_loss(logits, lab)
    loss.backward()
    return logits.detach(), loss.item(), {n: q.grad.detach().clone() for n, q in model.named_parameters() if q.grad is not None}


def _oracle_step(p, x, nb, lab, dtype, **kw):
    q = {k: v.to(dtype).clone().requires_grad_(True) for k, v in p.items()}
    logits = orc.forward_csa(x.to(dtype), nb.to(dtype), q, 1, **kw)
    loss = orc.masked_ce_loss(logits, lab)
    loss.backward()
    return logits.detach(), loss.item(), {k: v.grad for k, v in q.items()}


def _check_against_oracle(got, ref, n_grads=11):
    (logits, loss, grads), (r_logits, r_loss, r_grads) = got, ref
    assert (logits.cpu().double() - r_logits.double()).abs().max().item() < 1e-4
    assert abs(loss - r_loss) < 1e-5
    assert len(grads) == n_grads
    for n, g in grads.items():
        r = r_grads[n].double()
        assert (g.cpu().double() - r).abs().max().item() <= 1e-4 * r.abs().max().item(), n


@pytest.mark.parametrize("mode", [0, 1], ids=["fp32", "bf16x3"])
def test_config2_four_shapes_k2_against_the_oracle(mode):
    """BASELINE configs[1] at its full size (B = 4, K = 2, 10000 x 256): logits 1e-4, loss 1e-5, all 11 gradients 1e-4
    relative against the CPU oracle (fp32 closed form, itself pinned to the reference by G4 / G7)."""
    _set_mode(mode)
    B, K, n_cls = 4, 2, 39
    p, x, nb, lab = orc.conditioned_csa_case(np.random.default_rng(2002), B, K, 1, n_cls, 4.0, 2.0, 1.0)
    model = _module(p, n_cls, K).eval()
    got = _step(model, x.cuda(), nb.cuda(), lab.cuda())
    _check_against_oracle(got, _oracle_step(p, x, nb, lab, torch.float32))


@pytest.mark.parametrize("mode", [0, 1], ids=["fp32", "bf16x3"])
def test_config5_geometry_against_the_oracle(mode):
    """BASELINE configs[4]'s geometry (96 channels, d_k = d_v = 96, K = 4, blocks of 500) at 2 shapes x 5000 points, through
    CrossShapeAt: logits, loss and all 11 gradients against the float64 oracle."""
    _set_mode(mode)
    B, K, n_cls, C, N = 2, 4, 39, 96, 5000
    geo = dict(d_model=C, d_k=C, d_v=C, block=500, n_blocks=N // 500)
    p, x, nb, lab = orc.conditioned_csa_case(np.random.default_rng(5005), B, K, 1, n_cls, 4.0, 2.0, 1.0, n_points=N, d_model=C, d_k=C)
    model = _module(p, n_cls, K, **geo).eval()
    assert model.compatibility_q.weight.shape == (C, C) and model.logit.weight.shape == (n_cls, C, 1, 1)
    got = _step(model, x.cuda(), nb.cuda(), lab.cuda())
    ref = _oracle_step(p, x, nb, lab, torch.float64, d_k=C, d_v=C, block=500, n_blocks=N // 500)
    _check_against_oracle(got, ref)


def test_config3_full_size_against_the_oracle():
    """BASELINE configs[2] at its FULL size — 32 shapes x 10000 points x 256 channels, K = 3, one batch (the reference's B > 1
    compatibility-row bookkeeping couples the shapes of a batch, csa_models.py:220,227, so a smaller batch is another function)
    — in the default math mode against the CPU oracle (fp32 closed form, pinned to the reference by G4 / G7; ~30 GB and under a
    minute of host time): logits 1e-4, loss 1e-5, all 11 gradients 1e-4 relative.  Eval-mode arithmetic (the masks of train mode
    have no CPU counterpart at this size; test_config3_full_size_benched_step_is_checked holds train mode to the fp32 mode)."""
    _set_mode(1)
    B, K, n_cls = 32, 3, 39
    p, x, nb, lab = orc.conditioned_csa_case(np.random.default_rng(3003), B, K, 1, n_cls, 4.0, 2.0, 1.0)
    model = _module(p, n_cls, K).eval()
    got = _step(model, x.cuda(), nb.cuda(), lab.cuda())
    torch.cuda.empty_cache()
    _check_against_oracle(got, _oracle_step(p, x, nb, lab, torch.float32))


def test_config5_full_size_against_the_oracle():
    """BASELINE configs[4] at its FULL size — 8 shapes x 50000 points x 96 channels, K = 4, 100 blocks of 500 — in the default math
    mode against the CPU oracle (fp32 closed form): logits, loss, all 11 gradients."""
    _set_mode(1)
    B, K, n_cls, C, N = 8, 4, 39, 96, 50000
    geo = dict(d_model=C, d_k=C, d_v=C, block=500, n_blocks=N // 500)
    p, x, nb, lab = orc.conditioned_csa_case(np.random.default_rng(5055), B, K, 1, n_cls, 4.0, 2.0, 1.0, n_points=N, d_model=C, d_k=C)
    model = _module(p, n_cls, K, **geo).eval()
    got = _step(model, x.cuda(), nb.cuda(), lab.cuda())
    torch.cuda.empty_cache()
    _check_against_oracle(got, _oracle_step(p, x, nb, lab, torch.float32, d_k=C, d_v=C, block=500, n_blocks=N // 500))


def _full_size_properties(B, K, N, C, T, seed, tol_loss=1e-5, tol_grad=1e-4):
    """A full-size step where the CPU oracle is out of reach: the two parity-bearing math modes must agree with each other
    (eval mode: loss to tol_loss, every gradient to tol_grad relative; train mode with the same mask seeds likewise), eval
    mode must be bitwise repeatable, everything finite."""
    n_cls = 39
    geo = dict(d_model=C, d_k=C, d_v=C, block=T, n_blocks=N // T)
    rng = np.random.default_rng(seed)
    p = orc.make_params(rng, 1, d_model=C, d_k=C, d_v=C, n_cls=n_cls, csa=True)
    p["attention.fc.weight"] = p["attention.fc.weight"] * 4.0
    x = torch.from_numpy(rng.standard_normal((B, C, N, 1)).astype(np.float32)).cuda()
    x += torch.from_numpy(rng.standard_normal((B, C, 1, 1)).astype(np.float32)).cuda()
    nb = torch.empty((B, K + 1, C, N, 1), device="cuda")
    nb[:, 0] = x
    for k in range(K):
        nb[:, k + 1] = torch.from_numpy(rng.standard_normal((B, C, N, 1)).astype(np.float32)).cuda()
        nb[:, k + 1] += torch.from_numpy(rng.standard_normal((B, C, 1, 1)).astype(np.float32)).cuda()
    lab = orc.synth_labels(rng, B, N, n_cls).cuda()
    model = _module(p, n_cls, K, **geo)
    res = {}
    for mode in (0, 1):
        _set_mode(mode)
        model.eval()
        res[mode, "eval"] = _step(model, x, nb, lab)
        if mode == 1:
            again = _step(model, x, nb, lab)
            assert torch.equal(again[0], res[mode, "eval"][0]) and again[1] == res[mode, "eval"][1]      # eval: bitwise repeatable
            assert all(torch.equal(again[2][n], g) for n, g in res[mode, "eval"][2].items())
        model.train()
        torch.manual_seed(77)                                   # same mask seeds in both modes: the masks do not depend on the mode
        res[mode, "train"] = _step(model, x, nb, lab)
    for phase in ("eval", "train"):
        (l0, s0, g0), (l1, s1, g1) = res[0, phase], res[1, phase]
        assert torch.isfinite(l1).all() and np.isfinite(s1)
        assert (l0 - l1).abs().max().item() < 1e-4, phase
        assert abs(s0 - s1) < tol_loss, (phase, s0, s1)
        assert len(g0) == len(g1) == 11
        for n in g0:
            assert torch.isfinite(g1[n]).all()
            assert (g0[n] - g1[n]).abs().max().item() <= tol_grad * g0[n].abs().max().item(), (phase, n)
    assert abs(res[1, "train"][1] - res[1, "eval"][1]) > 1e-6          # dropout was live
    return res


def test_config5_full_size_properties():
    """BASELINE configs[4] at full size: 8 shapes x 50000 points x 96 channels, K = 4, 100 blocks of 500."""
    _full_size_properties(B=8, K=4, N=50000, C=96, T=500, seed=5050)


def test_config3_full_size_benched_step_is_checked():
    """The step bench.py times (BASELINE configs[2]: 32 x 10000 x 256, K = 3, train mode) — loss and every gradient of the
    bf16x3 mode against the exact-fp32 mode at full size, same dropout masks."""
    _full_size_properties(B=32, K=3, N=10000, C=256, T=500, seed=3030)


def _named_mode_report(B, K, N, C, T, seed, mode, name):
    """A BASELINE configuration in the reduced-precision arithmetic it NAMES (configs[1]: bf16, configs[4]: "fp16 MFMA"), at full
    size, against the same step in exact fp32 with the same dropout masks: finite everywhere, the error printed, and held to
    loose bounds (these modes are outside the 1e-4 contract; tests/test_gpu_lowprec.py reports them per operation).  The
    single-product modes run the score-recomputing data flows where they are the default (d <= 128), so this is also their
    full-size check."""
    n_cls = 39
    geo = dict(d_model=C, d_k=C, d_v=C, block=T, n_blocks=N // T)
    rng = np.random.default_rng(seed)
    p = orc.make_params(rng, 1, d_model=C, d_k=C, d_v=C, n_cls=n_cls, csa=True)
    p["attention.fc.weight"] = p["attention.fc.weight"] * 4.0
    x = torch.from_numpy(rng.standard_normal((B, C, N, 1)).astype(np.float32)).cuda()
    x += torch.from_numpy(rng.standard_normal((B, C, 1, 1)).astype(np.float32)).cuda()
    nb = torch.empty((B, K + 1, C, N, 1), device="cuda")
    nb[:, 0] = x
    for k in range(K):
        nb[:, k + 1] = torch.from_numpy(rng.standard_normal((B, C, N, 1)).astype(np.float32)).cuda()
        nb[:, k + 1] += torch.from_numpy(rng.standard_normal((B, C, 1, 1)).astype(np.float32)).cuda()
    lab = orc.synth_labels(rng, B, N, n_cls).cuda()
    model = _module(p, n_cls, K, **geo).train()
    res = {}
    for m in (0, mode):
        _set_mode(m)
        torch.manual_seed(123)                                  # the masks are functions of (seed, position), not of the mode
        res[m] = _step(model, x, nb, lab)
    (l0, s0, g0), (l1, s1, g1) = res[0], res[mode]
    assert torch.isfinite(l1).all() and np.isfinite(s1) and len(g1) == 11
    e_logit = (l0 - l1).abs().max().item()
    e_grad = {n: ((g0[n] - g1[n]).abs().max() / g0[n].abs().max()).item() for n in g0}
    worst = max(e_grad, key=e_grad.get)
    print(f"[named mode] {name}: max |dlogit| {e_logit:.2e}, |dloss| {abs(s0 - s1):.2e}, worst gradient {worst} {e_grad[worst]:.2e} (relative to its max)")
    assert e_logit < 5e-2 and abs(s0 - s1) < 2e-3
    for n, g in g1.items():
        assert torch.isfinite(g).all() and e_grad[n] < 3e-2, (n, e_grad[n])


def test_config2_full_size_in_bf16():
    """BASELINE configs[1] as named: CSA K = 2, 4 shapes x 10000 points x 256 channels, bf16."""
    _named_mode_report(B=4, K=2, N=10000, C=256, T=500, seed=2020, mode=2, name="config 2, bf16")


def test_config5_full_size_in_fp16():
    """BASELINE configs[4] as named: 8 shapes x 50000 points x 96 channels, K = 4, fp16 MFMA (forward; backward in bf16)."""
    _named_mode_report(B=8, K=4, N=50000, C=96, T=500, seed=5151, mode=3, name="config 5, fp16")
