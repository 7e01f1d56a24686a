"""Module-level parity on the MI355X: the drop-in classes of csn_amd.csa_models against (a) the CPU oracle on the
same seeded inputs and (b) the committed golden vectors that the reference itself produced
(tests/golden/make_golden.py).  Tolerance: the 1e-4 fp32 contract of BASELINE.json's north_star
(absolute on the O(1) LayerNorm-ed outputs / logits, relative on gradients)."""
import os

import numpy as np
import pytest
import torch

from oracle import csa_oracle as orc

pytestmark = pytest.mark.gpu
ROW_STRIDE = 97
ATOL = 1e-4


# the whole module-level contract (1e-4 on outputs, reference goldens, bit-exact kNN indices) is checked in both
# arithmetic modes of the contractions: exact fp32 matrix cores and bf16x3
@pytest.fixture(autouse=True, params=[0, 1], ids=["fp32", "bf16x3"])
def math_mode(request):
    from csn_amd import _lib
    _lib.build()
    _lib.check(_lib.lib().csn_set_math_mode(request.param))
    yield request.param
    _lib.lib().csn_set_math_mode(1)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _model(kind, p, H, n_cls, K=None):
    from csn_amd.csa_models import get_model
    m = get_model(kind, n_cls, H, K)
    missing, unexpected = m.load_state_dict(p, strict=False)
    assert not unexpected and all(k.startswith("fc_1.") for k in missing)
    return m.cuda().eval()


def test_state_dict_keys_match_reference_contract():
    from csn_amd.csa_models import get_model
    m = get_model("csa", 39, 8, 3)
    keys = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    want = {
        "fc_1.0.0.weight": (256, 928, 1, 1), "fc_1.0.1.weight": (256,), "fc_1.0.1.bias": (256,),
        "fc_1.0.1.running_mean": (256,), "fc_1.0.1.running_var": (256,), "fc_1.0.1.num_batches_tracked": (),
        "logit.weight": (39, 256, 1, 1), "attention.w_qs.weight": (2048, 256), "attention.w_ks.weight": (2048, 256),
        "attention.w_vs.weight": (2048, 256), "attention.fc.weight": (256, 2048), "attention.norm.weight": (256,),
        "attention.norm.bias": (256,), "compatibility_q.weight": (256, 256), "compatibility_q.bias": (256,),
        "compatibility_k.weight": (256, 256), "compatibility_k.bias": (256,),
    }
    assert keys == want
    ssa = get_model("ssa", 4, 1)
    assert set(ssa.state_dict()) == {k for k in want if not k.startswith("compatibility")}
    with pytest.raises(AttributeError):
        get_model("nope", 4, 1)
    # the two d_model = 928 factories (csa_models.py:406-409, 416-419): a 928-wide attention beside the 256-wide fc_1 / logit /
    # compatibility head — shapes as the reference builds them (probed by importing it; restated here as numbers)
    import csn_amd.csa_models as M
    assert {"backbone_ssa_fc_logit", "backbone_csa_fc_logit", "backbone_fc_ssa_logit", "backbone_fc_csa_logit"} <= set(M.__all__)
    want928 = dict(want)
    want928.update({"logit.weight": (39, 256, 1, 1), "attention.w_qs.weight": (2048, 928), "attention.w_ks.weight": (2048, 928),
                    "attention.w_vs.weight": (2048, 928), "attention.fc.weight": (928, 2048), "attention.norm.weight": (928,),
                    "attention.norm.bias": (928,)})
    m = M.backbone_csa_fc_logit(39, 8, 3)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == want928
    assert m.after_fc is False and m.attention_type == "csa" and m.K == 3
    m = M.backbone_ssa_fc_logit(39, 8)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: v for k, v in want928.items() if not k.startswith("compat")}
    assert m.after_fc is False and m.attention_type == "ssa"


def test_g2_self_attention_config1_shapes(golden_dir):
    """BASELINE config 1 family: unchunked self-attention at N in {500, 512, 2048}, C in {256, 128, 96}."""
    from csn_amd.csa_models import MultiHeadAttention
    g = _load(golden_dir, "g2_self_attention")
    for i in range(5):
        N, C, H, seed = (int(v) for v in g[f"g2_{i}_cfg"])
        rng = np.random.default_rng(seed)
        p = orc.make_params(rng, H, d_model=C, d_k=C, d_v=C, csa=False)
        x = orc.synth_points(rng, (1, C, N, 1))
        m = MultiHeadAttention(H, C, C, C).cuda().eval()
        m.load_state_dict({k[len("attention."):]: v for k, v in p.items() if k.startswith("attention.")})
        with torch.no_grad():
            y, _ = m.self_attention(x.cuda())
        y = y.cpu()
        assert y.shape == (1, N, C)
        assert np.abs(y[:, ::29].numpy() - g[f"g2_{i}_rows"]).max() < ATOL
        ref = orc.mha_full_self(x, p, H, C, C)
        assert (y - ref).abs().max().item() < ATOL


def test_g3_mha_forward_self_and_cross(golden_dir):
    from csn_amd.csa_models import MultiHeadAttention
    g = _load(golden_dir, "g3_mha_forward")
    for i in range(2):
        H, seed = (int(v) for v in g[f"g3_{i}_cfg"])
        rng = np.random.default_rng(seed)
        p = orc.make_params(rng, H, csa=False)
        xa = orc.synth_points(rng, (1, 256, 10000, 1))
        xb = orc.synth_points(rng, (1, 256, 10000, 1))
        m = MultiHeadAttention(H, 256, 256, 256).cuda().eval()
        m.load_state_dict({k[len("attention."):]: v for k, v in p.items() if k.startswith("attention.")})
        with torch.no_grad():
            xa_d, xb_d = xa.cuda(), xb.cuda()
            ys, _ = m(xa_d, xa_d, xa_d, "test")
            yc, attn = m(xa_d, xb_d, xb_d, "test", return_attn=True)
        assert ys.shape == (1, 10000, 256)
        assert np.abs(ys.cpu()[:, ::ROW_STRIDE].numpy() - g[f"g3_{i}_self_rows"]).max() < ATOL
        assert np.abs(yc.cpu()[:, ::ROW_STRIDE].numpy() - g[f"g3_{i}_cross_rows"]).max() < ATOL
        assert np.abs(attn.cpu()[0, :, 0].numpy() - g[f"g3_{i}_attn_last_row0"]).max() < 1e-6


def test_g8_unequal_head_widths_against_reference_goldens(golden_dir):
    """d_k != d_v and head widths without a kernel instance (csa_models.py:42 allows both): the module runs them at the next
    kernel width with zero-padded weights.  Outputs and the gradients of all six weight tensors against the REFERENCE's
    (golden set G8: (H, C, d_k, d_v) = (2, 256, 64, 128) and (1, 256, 256, 96) through the chunked cross call at N = 10000,
    (3, 96, 48, 80) through the unchunked self_attention), 1e-4, both math modes."""
    from csn_amd.csa_models import MultiHeadAttention
    g = _load(golden_dir, "g8_mha_unequal_head_widths")
    for i in range(3):
        H, C, dk, dv, N, seed, chunked = (int(v) for v in g[f"g8_{i}_cfg"])
        rng = np.random.default_rng(seed)
        p = orc.make_params(rng, H, d_model=C, d_k=dk, d_v=dv, csa=False)
        xa, xb = orc.synth_points(rng, (1, C, N, 1)), orc.synth_points(rng, (1, C, N, 1))
        gy = orc.synth_points(rng, (1, N, C))
        m = MultiHeadAttention(H, C, dk, dv).cuda().eval()
        m.load_state_dict({k[len("attention."):]: v for k, v in p.items() if k.startswith("attention.")})
        assert m.w_qs.weight.shape == (H * dk, C) and m.fc.weight.shape == (C, H * dv)      # the checkpoint layout is the reference's
        y = m(xa.cuda(), xb.cuda(), xb.cuda(), "test")[0] if chunked else m.self_attention(xa.cuda())[0]
        (y * gy.cuda()).sum().backward()
        assert np.abs(y.detach().cpu()[:, ::29].numpy() - g[f"g8_{i}_rows"]).max() < ATOL
        for name, prm in m.named_parameters():
            ref = g[f"g8_{i}_grad_{name}"]
            gr = prm.grad.detach().cpu()
            got = (gr if gr.numel() <= 10000 else gr.reshape(gr.shape[0], -1)[::5, ::7]).numpy()
            assert got.shape == ref.shape
            assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), (i, name, np.abs(got - ref).max() / np.abs(ref).max())
            st = g[f"g8_{i}_gstats_{name}"]
            assert abs(gr.double().norm().item() - st[1]) <= 1e-4 * st[1], (i, name)


def test_points_beyond_block_grid_are_ignored_and_short_inputs_raise():
    """csa_models.py:83-90: 20 blocks of 500 — N = 12000 yields 10000 rows, N = 2048 raises IndexError."""
    from csn_amd.csa_models import MultiHeadAttention
    rng = np.random.default_rng(5)
    p = orc.make_params(rng, 1, csa=False)
    m = MultiHeadAttention(1, 256, 256, 256).cuda().eval()
    m.load_state_dict({k[len("attention."):]: v for k, v in p.items() if k.startswith("attention.")})
    x = orc.synth_points(rng, (1, 256, 12000, 1))
    with torch.no_grad():
        xd = x.cuda()
        y, _ = m(xd, xd, xd, "test")
    assert y.shape == (1, 10000, 256)
    ref = orc.mha_blockdiag(x, x, x, p, 1)
    assert (y.cpu() - ref).abs().max().item() < ATOL
    with pytest.raises(IndexError):
        xs = x[:, :, :2048].cuda()
        m(xs, xs, xs, "test")


@pytest.mark.parametrize("N,T", [(360, 100), (1300, 500), (96, 64)])
def test_ragged_last_block(N, T):
    """n_blocks=None: every point takes part, the last block is short when N is not a multiple of the block (the reference's
    20 x 500 cannot: csa_models.py:83-90) — self and cross evaluations, gradients to weights and inputs, against the oracle's
    ragged closed form; and through CrossShapeAt."""
    from csn_amd.csa_models import MultiHeadAttention, get_model
    rng = np.random.default_rng(55)
    C, H, d = 64, 2, 32
    p = orc.make_params(rng, H, d_model=C, d_k=d, d_v=d, csa=False)
    m = MultiHeadAttention(H, C, d, d, block=T, n_blocks=None).cuda().eval()
    m.load_state_dict({k[len("attention."):]: v for k, v in p.items() if k.startswith("attention.")})
    xa, xb = (orc.synth_points(rng, (2, C, N, 1)) for _ in range(2))
    xa_d, xb_d = xa.cuda().requires_grad_(True), xb.cuda().requires_grad_(True)
    ys, _ = m(xa_d, xa_d, xa_d, "test")
    yc, attn = m(xa_d, xb_d, xb_d, "test", return_attn=True)
    assert ys.shape == (2, N, C) and attn.shape == (2, H, N - (N - 1) // T * T, N - (N - 1) // T * T)
    gs, gc = (torch.from_numpy(rng.standard_normal((2, N, C)).astype(np.float32)) for _ in range(2))
    ((ys * gs.cuda()).sum() + (yc * gc.cuda()).sum()).backward()
    p64 = {k: v.double().requires_grad_(True) for k, v in p.items() if k.startswith("attention.")}
    xa64, xb64 = xa.double().requires_grad_(True), xb.double().requires_grad_(True)
    rs = orc.mha_blockdiag(xa64, xa64, xa64, p64, H, d_k=d, d_v=d, block=T, n_blocks=None)
    rc, rattn = orc.mha_blockdiag(xa64, xb64, xb64, p64, H, d_k=d, d_v=d, block=T, n_blocks=None, return_attn=True)
    ((rs * gs.double()).sum() + (rc * gc.double()).sum()).backward()
    assert (ys.detach().cpu().double() - rs.detach()).abs().max().item() < ATOL
    assert (yc.detach().cpu().double() - rc.detach()).abs().max().item() < ATOL
    assert (attn.cpu().double() - rattn[-1].detach()).abs().max().item() < 1e-5
    rel = lambda got, ref: ((got.cpu().double() - ref).abs().max() / ref.abs().max()).item()
    assert rel(xa_d.grad, xa64.grad) < 1e-4
    assert rel(xb_d.grad, xb64.grad) < 1e-4
    for name, prm in m.named_parameters():
        assert rel(prm.grad, p64["attention." + name].grad) < 1e-4, name
    # the whole CSA module on a ragged geometry
    n_cls, K = 5, 2
    pc = orc.make_params(rng, 1, d_model=C, d_k=C, d_v=C, n_cls=n_cls, csa=True)
    x = orc.synth_points(rng, (2, C, N, 1))
    nb = orc.synth_points(rng, (2, K + 1, C, N, 1))
    nb[:, 0] = x
    model = get_model("csa", n_cls, 1, K, d_model=C, d_k=C, d_v=C, block=T, n_blocks=None)
    model.load_state_dict(pc, strict=False)
    model = model.cuda().eval()
    with torch.no_grad():
        logits = model(x.cuda(), "test", nb.cuda())
    ref = orc.forward_csa(x, nb, pc, 1, d_k=C, d_v=C, block=T, n_blocks=None)
    assert logits.shape == (2, n_cls, N, 1) and (logits.cpu() - ref).abs().max().item() < ATOL


def test_three_distinct_inputs():
    """MultiHeadAttention.forward(Q, K, V) with three different tensors (the signature allows it, csa_models.py:81)."""
    from csn_amd.csa_models import MultiHeadAttention
    rng = np.random.default_rng(15)
    C, H, N, T = 64, 2, 200, 100
    p = orc.make_params(rng, H, d_model=C, d_k=32, d_v=32, csa=False)
    m = MultiHeadAttention(H, C, 32, 32, block=T, n_blocks=2).cuda().eval()
    m.load_state_dict({k[len("attention."):]: v for k, v in p.items() if k.startswith("attention.")})
    xq, xk, xv = (orc.synth_points(rng, (2, C, N, 1)) for _ in range(3))
    for t in (xq, xk, xv):
        t.requires_grad_(True)
    y, _ = m(xq.cuda(), xk.cuda(), xv.cuda(), "test")
    ref = orc.mha_blockdiag(xq, xk, xv, p, H, d_k=32, d_v=32, block=T, n_blocks=2)
    assert (y.detach().cpu() - ref.detach()).abs().max().item() < ATOL


def test_config5_channel_count_96_forward_backward():
    """BASELINE config 5 geometry (C = D = 96, blocks of 500): self and cross evaluations with weight gradients against the
    oracle's closed form — in bf16x3 mode this drives the K/V tile planes with a head dimension that does not fill the
    staging threads evenly (768 pieces on 512 threads)."""
    from csn_amd.csa_models import MultiHeadAttention
    rng = np.random.default_rng(16)
    C, H, T, nb = 96, 1, 500, 4
    N = T * nb
    p = orc.make_params(rng, H, d_model=C, d_k=C, d_v=C, csa=False)
    m = MultiHeadAttention(H, C, C, C, block=T, n_blocks=nb).cuda().eval()
    m.load_state_dict({k[len("attention."):]: v for k, v in p.items() if k.startswith("attention.")})
    xa, xb = (orc.synth_points(rng, (2, C, N, 1)) for _ in range(2))
    ys, _ = m(xa.cuda(), xa.cuda(), xa.cuda(), "test")
    yc, _ = m(xa.cuda(), xb.cuda(), xb.cuda(), "test")
    gs, gc = (torch.from_numpy(rng.standard_normal((2, N, C)).astype(np.float32)) for _ in range(2))
    ((ys * gs.cuda()).sum() + (yc * gc.cuda()).sum()).backward()
    p64 = {k: v.double().requires_grad_(True) for k, v in p.items() if k.startswith("attention.")}
    rs = orc.mha_blockdiag(xa.double(), xa.double(), xa.double(), p64, H, d_k=C, d_v=C, block=T, n_blocks=nb)
    rc = orc.mha_blockdiag(xa.double(), xb.double(), xb.double(), p64, H, d_k=C, d_v=C, block=T, n_blocks=nb)
    ((rs * gs.double()).sum() + (rc * gc.double()).sum()).backward()
    assert (ys.detach().cpu().double() - rs.detach()).abs().max().item() < ATOL
    assert (yc.detach().cpu().double() - rc.detach()).abs().max().item() < ATOL
    for name, prm in m.named_parameters():
        ref = p64["attention." + name].grad
        assert ((prm.grad.cpu().double() - ref).abs().max() / ref.abs().max()).item() < 2e-4, name


class _Ready:
    """stands in for csn_amd.sharding.PendingStack with the exchange already complete"""

    def __init__(self, stack):
        self.stack = stack

    def wait(self):
        return self.stack


def test_overlapped_path_matches_single_call():
    """CrossShapeAt fed with a pending neighbour stack (the multi-GPU path: self-attention of the own shapes first, the
    rest after the exchange) reproduces the single-call path: logits, loss and all 11 weight gradients."""
    from csn_amd.csa_models import get_model
    rng = np.random.default_rng(31)
    B, K, H, n_cls = 2, 3, 1, 7
    # well-conditioned inputs (oracle.conditioned_csa_case): the compatibility-head gradients are held to the same 2e-4
    p, x, nb, lab = orc.conditioned_csa_case(rng, B, K, H, n_cls, 4.0, 3.0, 1.0)
    x, nb, lab = x.cuda(), nb.cuda().contiguous(), lab.cuda()
    outs = []
    for overlapped in (False, True):
        m = get_model("csa", n_cls, H, K)
        m.load_state_dict(p, strict=False)
        m = m.cuda().eval()
        logits = m(x, "test", _Ready(nb) if overlapped else nb)
        loss = orc.masked_ce_loss(logits, lab)
        loss.backward()
        outs.append((logits.detach(), loss.item(), {n: q.grad.clone() for n, q in m.named_parameters() if q.grad is not None}))
    (l0, s0, g0), (l1, s1, g1) = outs
    assert (l0 - l1).abs().max().item() < 2e-5 and abs(s0 - s1) < 1e-5
    assert set(g0) == set(g1) and len(g0) == 11
    for n in g0:
        scale = g0[n].abs().max().item()
        tol_n = 2e-4
        assert (g0[n] - g1[n]).abs().max().item() <= tol_n * scale + 1e-9, n
    # train mode: both dropouts live, masks drawn in a different order -> statistically equal, finite, different
    m = get_model("csa", n_cls, H, K)
    m.load_state_dict(p, strict=False)
    m = m.cuda().train()
    torch.manual_seed(3)
    lt = m(x, "train", _Ready(nb))
    assert torch.isfinite(lt).all() and (lt - l0).abs().mean().item() < 0.2 and (lt - l0).abs().max().item() > 1e-4


def test_host_neighbour_stack_equals_device_stack():
    """The reference's loader hands the neighbour stack over on the CPU (csa_training.py:198-202, csa_models.py:216): pageable
    or pinned, it crosses PCIe in one side-stream transfer under the self-attention of the query shapes and must give the
    logits, loss and 11 gradients of the device-resident call — with and without a trusted slot 0."""
    from csn_amd.csa_models import get_model
    rng = np.random.default_rng(53)
    B, K, H, n_cls = 2, 3, 1, 7
    p, x, nb, lab = orc.conditioned_csa_case(rng, B, K, H, n_cls, 4.0, 3.0, 1.0)
    xd, labd = x.cuda(), lab.cuda()

    def run(stack, trust):
        m = get_model("csa", n_cls, H, K)
        m.load_state_dict(p, strict=False)
        m = m.cuda().eval()
        m.trust_neighbor_slot0 = trust
        logits = m(xd, "test", stack)
        loss = orc.masked_ce_loss(logits, labd)
        loss.backward()
        torch.cuda.synchronize()
        return logits.detach(), loss.item(), {n: q.grad.clone() for n, q in m.named_parameters() if q.grad is not None}

    l0, s0, g0 = run(nb.cuda().contiguous(), False)
    scrambled = nb.clone()
    scrambled[:, 0] = 7.0                                     # slot 0 is NOT trusted by default: the module puts x there itself
    for stack, trust in ((nb.clone(), False), (nb.clone().pin_memory(), True), (scrambled, False)):
        l1, s1, g1 = run(stack, trust)
        assert (l0 - l1).abs().max().item() < 2e-5 and abs(s0 - s1) < 1e-5
        assert set(g0) == set(g1) and len(g0) == 11
        for n in g0:
            assert (g0[n] - g1[n]).abs().max().item() <= 2e-4 * g0[n].abs().max().item() + 1e-9, n


def test_fused_data_flow_equals_the_unfused_one():
    """The passes the step no longer makes (pooled sums from the out-projection epilogue, the mix gradient rebuilt inside the
    LayerNorm backward, dQ / dK / dV accumulated per slot in registers) against the plain forms of the same arithmetic
    (row-sum pass, per-evaluation gradient maps, one read-modify-write launch per colour): logits, loss and all 11 gradients."""
    from csn_amd import functional as CF
    from csn_amd.csa_models import get_model
    rng = np.random.default_rng(41)
    B, K, H, n_cls = 2, 3, 1, 7
    # well-conditioned inputs (oracle.conditioned_csa_case): the compatibility-head gradients are held to the same 2e-4
    p, x, nb, lab = orc.conditioned_csa_case(rng, B, K, H, n_cls, 4.0, 3.0, 1.0)
    x, nb, lab = x.cuda(), nb.cuda().contiguous(), lab.cuda()
    outs = []
    from csn_amd import tuning
    for fused in (True, False):
        m = get_model("csa", n_cls, H, K)
        m.load_state_dict(p, strict=False)
        m = m.cuda().eval()
        with tuning.override(fused_point_sums=fused, link_mix=fused, grouped_dkv=fused, grouped_dq=fused):
            logits = m(x, "test", nb)
            loss = orc.masked_ce_loss(logits, lab)
            loss.backward()
        outs.append((logits.detach(), loss.item(), {n: q.grad.clone() for n, q in m.named_parameters() if q.grad is not None}))
    (l0, s0, g0), (l1, s1, g1) = outs
    assert (l0 - l1).abs().max().item() < 2e-5 and abs(s0 - s1) < 1e-5
    assert set(g0) == set(g1) and len(g0) == 11
    for n in g0:
        scale = g0[n].abs().max().item()
        tol_n = 2e-4
        assert (g0[n] - g1[n]).abs().max().item() <= tol_n * scale + 1e-9, n


def _grad_check(model, g, key, expect):
    seen = 0
    for name, prm in model.named_parameters():
        if f"{key}_nograd_{name}" in g:
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, name
            continue
        ref = g[f"{key}_grad_{name}"]
        gr = prm.grad.detach().cpu()
        got = gr.numpy() if gr.numel() <= 10000 else gr.reshape(gr.shape[0], -1)[::17, ::13].contiguous().numpy()
        scale = np.abs(ref).max()
        # 1e-4 relative, plus the reference's own fp32 rounding noise on this tensor (its deviation from the float64
        # oracle, recorded by make_golden.py): the compatibility-head gradients are ~1e-7 differences of O(1) sums
        # (a 256-long MFMA fp32 accumulation chain is a plain fmaf chain: its rounding error is a few times that of
        #  the blocked CPU GEMM the reference ran on, hence the factor 10 on the noise term)
        # ... and ONLY for those four tensors: every other tensor is held to the plain 1e-4 (the compatibility head itself is
        # held to 1e-4 on the well-conditioned G7 cases, test_g7_conditioned_csa_holds_all_11_gradients_to_1e4)
        noise = float(g[f"{key}_gnoise_{name}"][0]) if name.startswith("compatibility") else 0.0
        assert np.abs(got - ref).max() <= 1e-4 * scale + 10.0 * noise, (name, np.abs(got - ref).max(), scale, noise)
        st = g[f"{key}_gstats_{name}"]
        assert abs(gr.double().norm().item() - st[1]) <= 1e-4 * st[1] + 10.0 * noise * np.sqrt(gr.numel()), name
        seen += 1
    assert seen == expect


def test_g4_csa_forward_backward_against_reference_goldens(golden_dir):
    g = _load(golden_dir, "g4_csa")
    for i in range(3):
        B, K, H, n_cls, seed = (int(v) for v in g[f"g4_{i}_cfg"])
        rng = np.random.default_rng(seed)
        p = orc.make_params(rng, H, n_cls=n_cls, csa=True)
        x = orc.synth_points(rng, (B, 256, 10000, 1))
        nb = orc.synth_points(rng, (B, K + 1, 256, 10000, 1))
        nb[:, 0] = x
        lab = orc.synth_labels(rng, B, 10000, n_cls)
        model = _model("csa", p, H, n_cls, K)
        logits = model(x.cuda(), "test", nb)                 # neighbours on the CPU, as csa_training.py:198-202 passes them
        assert logits.shape == (B, n_cls, 10000, 1)
        loss = orc.masked_ce_loss(logits, lab.cuda())
        loss.backward()
        rows = logits.detach().cpu().squeeze(-1).permute(0, 2, 1)[:, ::ROW_STRIDE].numpy()
        assert np.abs(rows - g[f"g4_{i}_logit_rows"]).max() < ATOL
        assert abs(loss.item() - g[f"g4_{i}_loss"][0]) < 1e-5
        with torch.no_grad():
            feats, comp, _ = model._csa_cm(x.cuda(), nb.cuda(), return_parts=True)
        assert np.abs(feats.cpu().permute(0, 2, 1)[:, ::ROW_STRIDE].numpy() - g[f"g4_{i}_feat_rows"]).max() < ATOL
        assert np.abs(comp.cpu().numpy() - g[f"g4_{i}_comp_oracle"]).max() < 1e-5
        _grad_check(model, g, f"g4_{i}", 11)


def test_g7_conditioned_csa_holds_all_11_gradients_to_1e4(golden_dir):
    """The compatibility head on a WELL-CONDITIONED problem (oracle.conditioned_csa_case: per-shape channel offsets, scaled
    fc / w_qs — the reference's own fp32 noise on these gradients is ~1e-6 relative, recorded in the goldens): every one of
    the 11 trained tensors, compatibility_{q,k}.{weight,bias} included, within 1e-4 relative of the reference's gradients,
    no noise allowance, in both math modes; and within 1e-4 of the float64 oracle's as well.  Case 2 is the geometry of the
    published checkpoint (8 heads, K = 4: get_csa_pred.py:35-36)."""
    g = _load(golden_dir, "g7_csa_conditioned")
    for i in range(3):
        B, K, H, n_cls, seed = (int(v) for v in g[f"g7_{i}_cfg"])
        fc_s, q_s, off = (float(v) for v in g[f"g7_{i}_scales"])
        p, x, nb, lab = orc.conditioned_csa_case(np.random.default_rng(seed), B, K, H, n_cls, fc_s, q_s, off)
        model = _model("csa", p, H, n_cls, K)
        logits = model(x.cuda(), "test", nb)
        loss = orc.masked_ce_loss(logits, lab.cuda())
        loss.backward()
        rows = logits.detach().cpu().squeeze(-1).permute(0, 2, 1)[:, ::ROW_STRIDE].numpy()
        assert np.abs(rows - g[f"g7_{i}_logit_rows"]).max() < ATOL
        assert abs(loss.item() - g[f"g7_{i}_loss"][0]) < 1e-5
        with torch.no_grad():
            _, comp, _ = model._csa_cm(x.cuda(), nb.cuda(), return_parts=True)
        assert np.abs(comp.cpu().numpy() - g[f"g7_{i}_comp_oracle"]).max() < 1e-5
        seen = 0
        for name, prm in model.named_parameters():
            if name.startswith("fc_1"):
                assert prm.grad is None
                continue
            ref = g[f"g7_{i}_grad_{name}"]
            gr = prm.grad.detach().cpu()
            got = gr.numpy() if gr.numel() <= 10000 else gr.reshape(gr.shape[0], -1)[::17, ::13].contiguous().numpy()
            scale = np.abs(ref).max()
            assert np.abs(got - ref).max() <= 1e-4 * scale, (name, np.abs(got - ref).max() / scale)
            st = g[f"g7_{i}_gstats_{name}"]
            assert abs(gr.double().norm().item() - st[1]) <= 1e-4 * st[1], name
            seen += 1
        assert seen == 11


def test_g5_ssa_forward_backward_against_reference_goldens(golden_dir):
    g = _load(golden_dir, "g5_ssa")
    for i in range(2):
        B, H, n_cls, seed = (int(v) for v in g[f"g5_{i}_cfg"])
        rng = np.random.default_rng(seed)
        p = orc.make_params(rng, H, n_cls=n_cls, csa=False)
        x = orc.synth_points(rng, (B, 256, 10000, 1))
        lab = orc.synth_labels(rng, B, 10000, n_cls)
        model = _model("ssa", p, H, n_cls)
        logits = model(x.cuda(), "train")
        loss = orc.masked_ce_loss(logits, lab.cuda())
        loss.backward()
        rows = logits.detach().cpu().squeeze(-1).permute(0, 2, 1)[:, ::ROW_STRIDE].numpy()
        assert np.abs(rows - g[f"g5_{i}_logit_rows"]).max() < ATOL
        assert abs(loss.item() - g[f"g5_{i}_loss"][0]) < 1e-5
        _grad_check(model, g, f"g5_{i}", 7)


def test_g10_after_fc_false_against_reference_goldens(golden_dir):
    """CrossShapeAt(..., after_fc=False) — the reference skips the attention and applies the logit layer to every point of the
    input (csa_models.py:191-202) — against the reference's own outputs: 'ssa' and 'csa', 10000 / 7001 / 12000 points."""
    from csn_amd.csa_models import CrossShapeAt
    g = _load(golden_dir, "g10_after_fc_false")
    for i in range(3):
        kind, B, N, H, K, n_cls, seed = (int(v) for v in g[f"g10_{i}_cfg"])
        rng = np.random.default_rng(seed)
        p = orc.make_params(rng, H, n_cls=n_cls, csa=kind == 1)
        x = orc.synth_points(rng, (B, 256, N, 1))
        lab = orc.synth_labels(rng, B, N, n_cls)
        nb = orc.synth_points(rng, (B, K + 1, 256, N, 1)) if kind == 1 else None
        if i == 1:          # through the factory: the 928-wide attention is never on the path (csa_models.py:197-202)
            from csn_amd.csa_models import backbone_csa_fc_logit
            model = backbone_csa_fc_logit(n_cls, H, K)
            model.load_state_dict({k: v for k, v in p.items() if not k.startswith("attention.")}, strict=False)
        else:
            model = CrossShapeAt(n_cls, 256, H, K or None, attention_type="csa" if kind == 1 else "ssa", after_fc=False)
            missing, unexpected = model.load_state_dict(p, strict=False)
            assert not unexpected and all(k.startswith("fc_1.") for k in missing)
        model = model.cuda().eval()
        logits = model(x.cuda(), "train", nb)
        assert tuple(logits.shape) == (B, n_cls, N, 1)
        loss = orc.masked_ce_loss(logits, lab.cuda())
        loss.backward()
        rows = logits.detach().cpu().squeeze(-1).permute(0, 2, 1)[:, ::ROW_STRIDE].numpy()
        assert np.abs(rows - g[f"g10_{i}_logit_rows"]).max() < ATOL
        assert abs(loss.item() - g[f"g10_{i}_loss"][0]) < 1e-5
        grads = {n: q.grad for n, q in model.named_parameters() if q.grad is not None}
        assert sorted(grads) == ["logit.weight"]
        ref = g[f"g10_{i}_grad_logit.weight"]
        assert np.abs(grads["logit.weight"].cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max()


def test_g11_big_category_graph_against_the_reference(golden_dir):
    """golden set G11 (the reference's own big-category branch, csa_training.py:138-155): `get_center_shape_indices` picks the
    reference's centre shapes (k-means over max-pooled HIP SSA features; sklearn of this image) and `get_knn_graph_big` returns
    the reference's candidate-relative index tables bit for bit, for the train and the test loader."""
    from csn_amd.csa_models import get_model
    g = _load(golden_dir, "g11_big_category_graph")
    S_train, S_test, n_centers, K, seed = (int(v) for v in g["g11_cfg"])
    rng = np.random.default_rng(seed)
    p = orc.make_params(rng, 1, n_cls=4, csa=False)
    train = orc.synth_clustered_shapes(rng, S_train, n_centers)
    test = orc.synth_clustered_shapes(rng, S_test, n_centers)
    model = get_model("ssa", 4, 1)
    missing, unexpected = model.load_state_dict(p, strict=False)
    assert not unexpected and all(k.startswith("fc_1.") for k in missing)
    model = model.cuda().eval()
    lab = torch.zeros((1, 10000), dtype=torch.int64)
    loader = lambda shapes: [(x[None].cuda(), lab) for x in shapes]            # batches of one, like DataLoader(FeaturesDataset, 1)
    centres = np.asarray(model.get_center_shape_indices(loader(train)))
    assert np.array_equal(np.sort(centres), g["g11_centres"])
    tr = model.get_knn_graph_big(loader(train), loader(train), centres.copy(), K).cpu()
    te = model.get_knn_graph_big(loader(test), loader(train), centres.copy(), K).cpu()
    assert tr.dtype == torch.int64 and np.array_equal(tr.numpy(), g["g11_train_graph"])
    assert np.array_equal(te.numpy(), g["g11_test_graph"])
    meas = model.get_retrieval_measure_big(loader(test), loader(train), centres.copy()).cpu().numpy()
    assert np.abs(meas - g["g11_test_measure"]).max() < 1e-5


def test_g6_knn_graph_indices_bit_exact(golden_dir):
    g = _load(golden_dir, "g6_retrieval")
    from csn_amd.csa_models import get_model
    model = get_model("ssa", 4, 1).cuda().eval()
    for i in range(2):
        S, N, K, seed = (int(v) for v in g[f"g6_{i}_cfg"])
        rng = np.random.default_rng(seed)
        f = orc.synth_clustered_feats(rng, S, N)
        r = model.get_retrieval_measure(f, f).cpu().numpy()
        assert np.abs(r - g[f"g6_{i}_measure"]).max() < 1e-5
        graph = model.get_knn_graph(f, f, K).cpu()
        assert graph.dtype == torch.int64
        assert np.array_equal(graph.numpy(), g[f"g6_{i}_graph"])


def test_input_gradients_when_requested():
    """The reference's features are constants, but the module is a normal autograd citizen: d loss / d x must match."""
    from csn_amd.csa_models import MultiHeadAttention
    rng = np.random.default_rng(9)
    C, H, N, T = 64, 2, 400, 100
    p = orc.make_params(rng, H, d_model=C, d_k=32, d_v=32, csa=False)
    m = MultiHeadAttention(H, C, 32, 32, block=T, n_blocks=4).cuda().eval()
    m.load_state_dict({k[len("attention."):]: v for k, v in p.items() if k.startswith("attention.")})
    xa = orc.synth_points(rng, (2, C, N, 1))
    xb = orc.synth_points(rng, (2, C, N, 1))
    wgt = orc.synth_points(rng, (2, N, C))
    xa_d, xb_d = xa.cuda().requires_grad_(True), xb.cuda().requires_grad_(True)
    y, _ = m(xa_d, xb_d, xb_d, "test")
    (y * wgt.cuda()).sum().backward()
    xa_r, xb_r = xa.double().requires_grad_(True), xb.double().requires_grad_(True)
    p64 = {k: v.double() for k, v in p.items()}
    yr = orc.mha_blockdiag(xa_r, xb_r, xb_r, p64, H, d_k=32, d_v=32, block=T, n_blocks=4)
    (yr * wgt.double()).sum().backward()
    assert (y.detach().cpu().double() - yr.detach()).abs().max().item() < ATOL
    for got, ref in ((xa_d.grad, xa_r.grad), (xb_d.grad, xb_r.grad)):
        assert ((got.cpu().double() - ref).abs().max() / ref.abs().max()).item() < 1e-4


def test_product_path_has_no_cpu_fallback():
    from csn_amd.csa_models import MultiHeadAttention
    from csn_amd import CsnError
    m = MultiHeadAttention(1, 32, 32, 32, block=36, n_blocks=1).eval()        # parameters left on the CPU
    x = torch.zeros(1, 32, 36, 1)
    with pytest.raises((CsnError, RuntimeError)):
        m(x.cuda(), x.cuda(), x.cuda(), "test")
