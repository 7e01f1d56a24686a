"""The fused masked cross-entropy pair (csn_amd/csrc/loss.hip behind csn_masked_ce_fwd_f32 / csn_masked_ce_bwd_f32) against the
oracle's restatement of the reference's loss (oracle/csa_oracle.py masked_ce_loss <- MID-FC/csa_training.py:94-108) in float64:
loss, accuracy, counted points and the gradient, on class-major logits that are a strided view of a wider buffer, with ragged
point counts, labels at the mask, and a batch with nothing to count."""
import numpy as np
import pytest
import torch

from oracle import csa_oracle as orc

pytestmark = pytest.mark.gpu


def _case(S, n_cls, N, pad_rows, seed, p_ignored=0.15, scale=3.0):
    rng = np.random.default_rng(seed)
    buf = torch.from_numpy((scale * rng.standard_normal((S, n_cls + pad_rows, N))).astype(np.float32))
    lab = torch.from_numpy(np.where(rng.random((S, N)) < p_ignored, 0, rng.integers(0, n_cls, size=(S, N))).astype(np.int64))
    return buf, lab


@pytest.mark.parametrize("S,n_cls,N,pad_rows", [(3, 39, 1000, 1), (32, 39, 10000, 1), (1, 5, 4, 0), (2, 50, 260, 2), (4, 2, 516, 0),
                                                (2, 39, 1001, 1), (3, 7, 3, 0), (1, 11, 10007, 0)])        # N % 4 != 0: the reference's loss takes any N
def test_loss_accuracy_and_gradient_against_the_oracle(S, n_cls, N, pad_rows):
    from csn_amd.functional import masked_cross_entropy
    buf, lab = _case(S, n_cls, N, pad_rows, seed=S + n_cls)
    bd = buf.cuda().requires_grad_(True)
    logits = bd[:, :n_cls]                                   # the logit layer pads its rows to a multiple of 4: a strided view
    loss, accu, count = masked_cross_entropy(logits.unsqueeze(-1), lab.cuda(), 0)
    g = torch.tensor(1.7, device="cuda")
    (loss * g).backward()
    torch.cuda.synchronize()
    ref_in = buf[:, :n_cls].double().requires_grad_(True)
    ref = orc.masked_ce_loss(ref_in.unsqueeze(-1), lab)
    (ref * 1.7).backward()
    keep = lab > 0
    assert count.item() == keep.sum().item()
    assert abs(loss.item() - ref.item()) < 2e-6 * max(1.0, abs(ref.item()))
    ref_acc = (buf[:, :n_cls].argmax(dim=1)[keep] == lab[keep]).double().mean().item()
    assert abs(accu.item() - ref_acc) < 1e-6
    dg = bd.grad.cpu()
    assert torch.all(dg[:, n_cls:] == 0)                      # the padding rows of the buffer get no gradient
    err = (dg[:, :n_cls].double() - ref_in.grad).abs().max().item()
    assert err < 2e-6 * ref_in.grad.abs().max().item() + 1e-12
    assert torch.all(dg[:, :n_cls].permute(0, 2, 1)[~keep] == 0)     # points at or below the mask: exactly zero


def test_it_is_reproducible_bit_for_bit_and_the_mask_is_a_threshold():
    from csn_amd.functional import masked_cross_entropy
    buf, lab = _case(8, 39, 5000, 1, seed=3)
    lab[lab == 1] = 2
    lab[0, :100] = 1                                          # mask = 1: labels 0 and 1 are dropped (csa_training.py:101)
    ld, ll = buf.cuda()[:, :39], lab.cuda()
    a = [t.item() for t in masked_cross_entropy(ld, ll, 1)]
    b = [t.item() for t in masked_cross_entropy(ld, ll, 1)]
    assert a == b
    ref = orc.masked_ce_loss(buf[:, :39].double().unsqueeze(-1), lab, mask=1)
    assert a[2] == (lab > 1).sum().item() and abs(a[0] - ref.item()) < 2e-6 * abs(ref.item())


def test_nothing_to_count_gives_nan_like_the_reference_and_a_zero_gradient_is_not_invented():
    from csn_amd.functional import masked_cross_entropy
    buf, _ = _case(2, 39, 512, 1, seed=9)
    lab = torch.zeros((2, 512), dtype=torch.int64)
    loss, accu, count = masked_cross_entropy(buf.cuda()[:, :39], lab.cuda(), 0)
    ref = orc.masked_ce_loss(buf[:, :39].unsqueeze(-1), lab)
    assert torch.isnan(ref) and torch.isnan(loss) and count.item() == 0


def test_the_training_module_routes_device_logits_through_it():
    from csn_amd import training
    buf, lab = _case(4, 39, 2000, 1, seed=11)
    calls = []
    from csn_amd import _lib
    _lib.set_call_hook(lambda name, phase: calls.append(name) if phase == "begin" else None)
    try:
        loss, accu = training.loss_functions_seg(buf.cuda()[:, :39].unsqueeze(-1), lab.cuda(), 39)
    finally:
        _lib.set_call_hook(None)
    assert "csn_masked_ce_fwd_f32" in calls
    r_loss, r_accu = training.loss_functions_seg(buf[:, :39].unsqueeze(-1), lab, 39)          # host tensors: the torch sequence
    assert abs(loss.item() - r_loss.item()) < 2e-6 * r_loss.item() and abs(accu.item() - r_accu.item()) < 1e-6


def test_wrong_types_are_refused():
    from csn_amd import _lib
    from csn_amd.functional import masked_cross_entropy
    buf, lab = _case(2, 39, 512, 1, seed=9)
    with pytest.raises(_lib.CsnError):
        masked_cross_entropy(buf.cuda()[:, :39].half(), lab.cuda(), 0)
    with pytest.raises(_lib.CsnError):
        masked_cross_entropy(buf.cuda()[:, :39], lab.cuda().int(), 0)


def test_the_reference_s_own_loss_and_gradients_through_the_fused_pair(golden_dir):
    """Golden set G7 holds the loss value and the 11 gradients the REFERENCE produced (its loss_functions_seg on its logits,
    tests/golden/make_golden.py).  The module's logits through the fused pair give that loss to 1e-5 and — the loss' gradient is
    where the backward starts — the reference's gradients to 1e-4, like the oracle's loss does in tests/test_gpu_module.py."""
    import os
    from csn_amd.csa_models import get_model
    from csn_amd.functional import masked_cross_entropy
    g = np.load(os.path.join(golden_dir, "g7_csa_conditioned.npz"))
    for i in (0, 1):
        B, K, H, n_cls, seed = (int(v) for v in g[f"g7_{i}_cfg"])
        fc_s, q_s, off = (float(v) for v in g[f"g7_{i}_scales"])
        p, x, nb, lab = orc.conditioned_csa_case(np.random.default_rng(seed), B, K, H, n_cls, fc_s, q_s, off)
        model = get_model("csa", n_cls, H, K)
        model.load_state_dict(p, strict=False)
        model = model.cuda().eval()
        logits = model(x.cuda(), "test", nb)
        loss, accu, count = masked_cross_entropy(logits, lab.cuda().long(), 0)
        loss.backward()
        assert abs(loss.item() - g[f"g7_{i}_loss"][0]) < 1e-5
        assert count.item() == (lab > 0).sum().item()
        seen = 0
        for name, prm in model.named_parameters():
            if name.startswith("fc_1"):
                continue
            ref = g[f"g7_{i}_grad_{name}"]
            gr = prm.grad.detach().cpu()
            got = gr.numpy() if gr.numel() <= 10000 else gr.reshape(gr.shape[0], -1)[::17, ::13].contiguous().numpy()
            assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), name
            seen += 1
        assert seen == 11


def test_any_point_count_and_a_view_that_is_not_16_byte_aligned():
    """csa_training.py:94-108 takes any N and any view; the backward's 16-byte form needs N, pitches % 4 == 0 and aligned rows, so
    other geometries run its one-point-per-thread form instead of failing inside autograd (round-4 advisor finding)."""
    from csn_amd.functional import masked_cross_entropy
    rng = np.random.default_rng(5)
    S, n_cls, N = 2, 39, 1001
    wide = torch.from_numpy((2.0 * rng.standard_normal((S, n_cls, N + 3))).astype(np.float32))
    lab = torch.from_numpy(np.where(rng.random((S, N)) < 0.2, 0, rng.integers(0, n_cls, size=(S, N))).astype(np.int64))
    wd = wide.cuda().requires_grad_(True)
    view = wd[:, :, 1:1 + N]                                  # rows start 4 bytes into a 16-byte unit, N % 4 == 1
    assert view.data_ptr() % 16 != 0
    loss, accu, count = masked_cross_entropy(view, lab.cuda(), 0)
    loss.backward()
    torch.cuda.synchronize()
    ref_in = wide[:, :, 1:1 + N].double().requires_grad_(True)
    ref = orc.masked_ce_loss(ref_in.unsqueeze(-1), lab)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 2e-6 * abs(ref.item()) and count.item() == (lab > 0).sum().item()
    g = wd.grad.cpu()
    assert torch.all(g[:, :, 0] == 0) and torch.all(g[:, :, 1 + N:] == 0)
    assert (g[:, :, 1:1 + N].double() - ref_in.grad).abs().max().item() < 2e-6 * ref_in.grad.abs().max().item()


def test_a_logit_of_minus_infinity_is_a_probability_of_zero_not_a_nan():
    """torch's log_softmax handles -inf logits (a masked class); the online log-sum-exp must too, also when -inf comes FIRST."""
    from csn_amd.functional import masked_cross_entropy
    buf, lab = _case(2, 6, 64, 0, seed=9)
    buf[:, 0, :] = -float("inf")                              # first class of every point
    buf[0, 3, ::2] = -float("inf")
    lab[lab == 0] = 1
    lab[(buf.gather(1, lab.unsqueeze(1)).squeeze(1) == -float("inf"))] = 2      # never the label itself (that loss IS inf)
    lab[(buf.gather(1, lab.unsqueeze(1)).squeeze(1) == -float("inf"))] = 4
    bd = buf.cuda().requires_grad_(True)
    loss, accu, count = masked_cross_entropy(bd, lab.cuda(), 0)
    loss.backward()
    torch.cuda.synchronize()
    ref_in = buf.double().requires_grad_(True)
    ref = orc.masked_ce_loss(ref_in.unsqueeze(-1), lab)
    ref.backward()
    assert np.isfinite(loss.item()) and abs(loss.item() - ref.item()) < 2e-6 * abs(ref.item())
    g = bd.grad.cpu()
    assert torch.isfinite(g).all() and torch.all(g[:, 0] == 0)
    assert (g.double() - ref_in.grad).abs().max().item() < 2e-6 * ref_in.grad.abs().max().item()
