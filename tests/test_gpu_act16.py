"""16-bit activation maps between the launches (math modes 2 and 3; include/csn_hip.h "16-BIT ACTIVATION MAPS"): the same
step with Qs / Ctx / xhat / dZ / dCtx exchanged as one 16-bit plane against the fp32 exchange.  Wherever a map is only a matrix
operand the products are the same bits (it was rounded to those 16 bits at staging anyway); delta, the LayerNorm backward
and the mix see the rounded maps, so the comparison is to the rounding of the mode, not bitwise."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from csn_amd import _lib
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    yield _lib
    _lib.lib().csn_set_math_mode(1)
    _lib.lib().csn_set_thread_math_mode(-1)
    _lib.lib().csn_set_thread_act16(0)


def _step(model, x, nbf, lab, seed):
    from oracle import csa_oracle as orc
    for prm in model.parameters():
        prm.grad = None
    torch.manual_seed(seed)
    logits = model(x, "train", nbf)
    loss = orc.masked_ce_loss(logits, lab)
    loss.backward()
    return logits.detach().clone(), loss.item(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}


GEOS = [dict(), dict(d_model=128, d_k=128, d_v=128, block=100, n_blocks=3), dict(d_model=96, d_k=96, d_v=96, block=500, n_blocks=2),
        dict(d_model=64, d_k=32, d_v=32, n_head=2, block=52, n_blocks=3),
        dict(d_model=128, d_k=128, d_v=128, block=100, n_blocks=None, n_pts=236),       # the row ends inside the last block (36 points)
        dict(d_model=256, d_k=32, d_v=32, n_head=8, block=300, n_blocks=None, n_pts=444)]   # 8 heads, short last block, 256 x 256 GEMM tiles


@pytest.mark.parametrize("geo", GEOS, ids=["d256", "d128", "d96", "2heads-d32-ragged-tiles", "d128-short-last-block", "8heads-short-last-block"])
@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("train", [False, True], ids=["eval", "train"])
def test_module_step_with_16bit_maps(L, mode, geo, train):
    """CrossShapeAt forward + masked CE + backward with and without the 16-bit exchange, same weights, inputs and masks."""
    from csn_amd import tuning
    from csn_amd.csa_models import get_model
    rng = np.random.default_rng(31)
    B, K, n_cls = 2, 2, 7
    C = geo.get("d_model", 256)
    N = geo.get("n_pts") or geo.get("block", 500) * geo.get("n_blocks", 20)
    torch.manual_seed(5)
    model = get_model("csa", n_cls, geo.get("n_head", 1), K, **{k: v for k, v in geo.items() if k not in ("n_head", "n_pts")}).cuda().train(train)
    L.check(L.lib().csn_set_math_mode({"bf16": 2, "fp16": 3}[mode]))
    off = torch.from_numpy(rng.standard_normal((B, K + 1, C, 1, 1)).astype(np.float32))
    nbf = (torch.from_numpy(rng.standard_normal((B, K + 1, C, N, 1)).astype(np.float32)) + 2.0 * off).cuda()
    x = nbf[:, 0].contiguous()
    lab = torch.from_numpy(rng.integers(0, n_cls, size=(B, N))).cuda()
    with tuning.override(act16=False):
        l0, s0, g0 = _step(model, x, nbf, lab, 9)
    with tuning.override(act16=True):
        l1, s1, g1 = _step(model, x, nbf, lab, 9)
    assert len(g0) == 11 and set(g0) == set(g1)
    # logits: the mix reads xhat rounded to fp16 (2^-11 relative per value)
    scale = l0.abs().max().item()
    assert (l0 - l1).abs().max().item() <= 2e-3 * scale
    assert abs(s0 - s1) <= 1e-3 * abs(s0)
    for n in g0:
        gs = g0[n].abs().max().item()
        err = (g0[n] - g1[n]).abs().max().item()
        assert err <= 3e-2 * gs, (n, err, gs)


def test_input_gradients_with_16bit_maps(L):
    """The same step with the neighbour stack requiring a gradient: dx = W^T dqkv reads the bf16 gradient maps, the residual
    branch stays fp32."""
    from csn_amd import tuning
    from csn_amd.csa_models import get_model
    from oracle import csa_oracle as orc
    rng = np.random.default_rng(32)
    B, K, n_cls, C, N = 2, 2, 5, 128, 300
    torch.manual_seed(6)
    model = get_model("csa", n_cls, 1, K, d_model=C, d_k=128, d_v=128, block=100, n_blocks=3).cuda().train(True)
    L.check(L.lib().csn_set_math_mode(2))
    nb0 = torch.from_numpy(rng.standard_normal((B, K + 1, C, N, 1)).astype(np.float32)).cuda()
    lab = torch.from_numpy(rng.integers(0, n_cls, size=(B, N))).cuda()
    grads = []
    for on in (False, True):
        nbf = nb0.clone().requires_grad_(True)
        with tuning.override(act16=on):
            torch.manual_seed(3)
            loss = orc.masked_ce_loss(model(nbf[:, 0], "train", nbf), lab)
            loss.backward()
        grads.append(nbf.grad.clone())
    scale = grads[0].abs().max().item()
    assert scale > 0 and (grads[0] - grads[1]).abs().max().item() <= 3e-2 * scale


class _Ready:
    """stands in for csn_amd.sharding.PendingStack with the exchange already complete"""

    def __init__(self, stack):
        self.stack = stack

    def wait(self):
        return self.stack


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_overlapped_path_with_16bit_maps(L, mode):
    """The multi-GPU form of the step (own shapes first, the rest after the exchange: two evaluation batches whose fp16 maps
    meet in one mix) on the 16-bit exchange against the single call on fp32 maps."""
    from csn_amd import tuning
    from csn_amd.csa_models import get_model
    from oracle import csa_oracle as orc
    rng = np.random.default_rng(33)
    B, K, H, n_cls = 2, 3, 1, 7
    p, x, nb, lab = orc.conditioned_csa_case(rng, B, K, H, n_cls, 4.0, 3.0, 1.0)
    x, nb, lab = x.cuda(), nb.cuda().contiguous(), lab.cuda()
    outs = []
    for overlapped, on in ((False, False), (True, True)):
        m = get_model("csa", n_cls, H, K, math=mode)
        m.load_state_dict(p, strict=False)
        m = m.cuda().eval()
        with tuning.override(act16=on):
            logits = m(x, "test", _Ready(nb) if overlapped else nb)
            loss = orc.masked_ce_loss(logits, lab)
            loss.backward()
        outs.append((logits.detach(), loss.item(), {n: q.grad.clone() for n, q in m.named_parameters() if q.grad is not None}))
    (l0, s0, g0), (l1, s1, g1) = outs
    assert (l0 - l1).abs().max().item() <= 2e-3 * l0.abs().max().item() and abs(s0 - s1) <= 1e-3 * abs(s0)
    assert set(g0) == set(g1) and len(g0) == 11
    for n in g0:
        assert (g0[n] - g1[n]).abs().max().item() <= 3e-2 * g0[n].abs().max().item(), n


def test_switch_is_scoped_to_the_thread(L):
    lib = L.lib()
    assert lib.csn_get_thread_act16() == 0
    L.check(lib.csn_set_thread_act16(2))
    assert lib.csn_get_thread_act16() == 2
    import threading
    seen = []
    t = threading.Thread(target=lambda: seen.append(lib.csn_get_thread_act16()))
    t.start(); t.join()
    assert seen == [0]
    assert lib.csn_set_thread_act16(3) == -1 and lib.csn_set_thread_act16(4) == -1 and lib.csn_set_thread_act16(7) == -1
    L.check(lib.csn_set_thread_act16(5))
    L.check(lib.csn_set_thread_act16(0))


def test_forward_refuses_a_type_that_is_not_its_modes(L):
    lib = L.lib()
    L.check(lib.csn_set_math_mode(2))
    L.check(lib.csn_set_thread_act16(2))                 # fp16 maps in a bf16 forward
    x = torch.zeros((1, 32, 64), device="cuda")
    w = torch.zeros((32, 32), device="cuda")
    xhat = torch.zeros((1, 32, 64), device="cuda")
    rstd = torch.zeros((1, 64), device="cuda")
    rc = lib.csn_outproj_ln_fwd_f32(x.data_ptr(), 32 * 64, w.data_ptr(), x.data_ptr(), 32 * 64, None, xhat.data_ptr(), 32 * 64,
                                    rstd.data_ptr(), 1, 32, 32, 64, 64, 1e-6, 0.0, 0, None, None, 0,
                                    torch.cuda.current_stream().cuda_stream)
    L.check(lib.csn_set_thread_act16(0))
    assert rc == -1


@pytest.mark.parametrize("S,R,C,NP", [(3, 40, 36, 300), (2, 256, 256, 1004), (1, 96, 96, 236), (2, 288, 96, 520)])
def test_projection_entry_points_on_16bit_maps(L, S, R, C, NP):
    """csn_project_f32 writing a bf16 map (out_split = 3) and reading one (+ 16), csn_project_wgrad_f32 contracting bf16
    gradient maps (flag 5), against float64 on the SAME rounded operands: sizes whose rows and contraction lengths end inside
    a 16-byte unit (NP % 8 == 4, C % 8 == 4), small and 256 x 256 tiles."""
    lib = L.lib()
    L.check(lib.csn_set_math_mode(2))
    rng = np.random.default_rng(S * 1000 + NP)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.from_numpy(rng.standard_normal((S, C, NP)).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((R, C)) / np.sqrt(C)).astype(np.float32)).cuda()
    bf = lambda t: t.bfloat16().double()
    # (a) out = w @ x as a bf16 map
    out16 = torch.full((S, R, NP), float("nan"), device="cuda", dtype=torch.bfloat16)
    L.check(lib.csn_project_f32(x.data_ptr(), C * NP, NP, w.data_ptr(), R, C, out16.data_ptr(), R * NP, NP, S, NP, 0, 1.0, 3, 0, st))
    ref = torch.einsum("rc,scn->srn", bf(w), bf(x))
    assert torch.isfinite(out16.float()).all()
    assert (out16.double() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item()          # one bf16 rounding of the result
    # (b) dx = w^T @ g with g a bf16 map as INPUT (out_split + 16), fp32 result
    g16 = torch.from_numpy(rng.standard_normal((S, R, NP)).astype(np.float32)).cuda().bfloat16()
    wt = w.t().contiguous()
    dx = torch.full((S, C, NP), float("nan"), device="cuda")
    L.check(lib.csn_project_f32(g16.data_ptr(), R * NP, NP, wt.data_ptr(), C, R, dx.data_ptr(), C * NP, NP, S, NP, 0, 1.0, 16, 0, st))
    ref = torch.einsum("cr,srn->scn", bf(wt), g16.double())
    assert (dx.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    # (c) dw = sum_s g[s] @ x[s]^T with g a bf16 map (flag 5), x fp32 (rounded to bf16 while staged)
    L.check(lib.csn_set_thread_act16(5))
    try:
        ws_n = lib.csn_wgrad_workspace_floats(R, C, S, NP)
        ws = torch.empty((ws_n,), device="cuda")
        dw = torch.full((R, C), float("nan"), device="cuda")
        L.check(lib.csn_project_wgrad_f32(g16.data_ptr(), R * NP, NP, x.data_ptr(), C * NP, NP, dw.data_ptr(), R, C, S, NP, 1.0, 0,
                                          ws.data_ptr(), ws_n, st))
    finally:
        L.check(lib.csn_set_thread_act16(0))
    ref = torch.einsum("srn,scn->rc", g16.double(), bf(x))
    assert (dw.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize("switch", ["fused_point_sums", "grouped_dq", "grouped_dkv", "link_mix", "keep_flow"])
def test_16bit_maps_with_each_data_flow_switch_off(L, switch):
    """The exchange under every other data-flow switch of csn_amd.tuning turned off: pooled sums by the streaming pass over the
    fp16 maps, dQ / dK / dV by one read-modify-write launch per colour (fp32 gradient maps then), an unlinked mix (no 16-bit maps at
    all: the switch must fall back, not fail), kept scores at a width that would take the flash flow."""
    from csn_amd import tuning
    from csn_amd.csa_models import get_model
    rng = np.random.default_rng(41)
    B, K, n_cls, C, N = 2, 2, 5, 128, 300
    torch.manual_seed(8)
    model = get_model("csa", n_cls, 1, K, d_model=C, d_k=128, d_v=128, block=100, n_blocks=3).cuda().train(True)
    L.check(L.lib().csn_set_math_mode(2))
    nbf = torch.from_numpy(rng.standard_normal((B, K + 1, C, N, 1)).astype(np.float32)).cuda()
    x = nbf[:, 0].contiguous()
    lab = torch.from_numpy(rng.integers(0, n_cls, size=(B, N))).cuda()
    off = {"keep_flow": dict(score_flow={1: tuning.KEEP_SCORES, 2: tuning.KEEP_SCORES})}.get(switch, {switch: False})
    with tuning.override(act16=False, **off):
        l0, s0, g0 = _step(model, x, nbf, lab, 4)
    with tuning.override(act16=True, **off):
        l1, s1, g1 = _step(model, x, nbf, lab, 4)
    assert (l0 - l1).abs().max().item() <= 2e-3 * l0.abs().max().item() and abs(s0 - s1) <= 1e-3 * abs(s0)
    for n in g0:
        assert (g0[n] - g1[n]).abs().max().item() <= 3e-2 * g0[n].abs().max().item(), n


@pytest.mark.parametrize("S,R,C,NP", [(3, 256, 256, 500), (2, 256, 256, 1004), (5, 96, 96, 36), (1, 260, 256, 10004)])
def test_16bit_operand_at_the_very_end_of_its_allocation(L, S, R, C, NP):
    """The fault of round 3 where it bit: a k-contiguous bf16 operand whose last row ends its allocation, contraction lengths
    with NP % 8 == 4 (the last 16-byte unit of every row straddles the end of the contraction).  The gradient maps are carved
    out of ONE pool so that they end exactly where a guard of bf16 NaNs begins: a unit fetched past the operand's window, or an
    upper half that is not cleared, puts a NaN into the product (NaN x 0 is NaN) — the result must be finite and right.  The
    window arithmetic itself is tested on the host (tests/test_cpu_window.py)."""
    lib = L.lib()
    L.check(lib.csn_set_math_mode(2))
    rng = np.random.default_rng(S * 7 + NP)
    st = torch.cuda.current_stream().cuda_stream
    n_el = S * R * NP
    pool = torch.full((n_el + 4096,), float("nan"), device="cuda", dtype=torch.bfloat16)      # operand + guard, one allocation
    g16 = pool[:n_el].view(S, R, NP)
    g16.copy_(torch.from_numpy(rng.standard_normal((S, R, NP)).astype(np.float32)).cuda())
    assert torch.isnan(pool[n_el:]).all() and g16.data_ptr() + 2 * n_el == pool[n_el:].data_ptr()
    x = torch.from_numpy(rng.standard_normal((S, C, NP)).astype(np.float32)).cuda()
    L.check(lib.csn_set_thread_act16(5))
    try:
        ws_n = lib.csn_wgrad_workspace_floats(R, C, S, NP)
        ws = torch.empty((ws_n,), device="cuda")
        dw = torch.full((R, C), float("nan"), device="cuda")
        L.check(lib.csn_project_wgrad_f32(g16.data_ptr(), R * NP, NP, x.data_ptr(), C * NP, NP, dw.data_ptr(), R, C, S, NP, 1.0, 0,
                                          ws.data_ptr(), ws_n, st))
    finally:
        L.check(lib.csn_set_thread_act16(0))
        L.check(lib.csn_set_math_mode(1))
    torch.cuda.synchronize()
    assert torch.isfinite(dw).all()                                    # no guard byte reached a product
    ref = torch.einsum("srn,scn->rc", g16.double(), x.bfloat16().double())
    assert (dw.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    assert torch.isnan(pool[n_el:]).all()                              # and nothing was written there either
