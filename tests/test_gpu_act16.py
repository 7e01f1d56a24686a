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
        dict(d_model=64, d_k=32, d_v=32, n_head=2, block=52, n_blocks=3)]


@pytest.mark.parametrize("geo", GEOS, ids=["d256", "d128", "d96", "2heads-d32-ragged-tiles"])
@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("train", [False, True], ids=["eval", "train"])
def test_module_step_with_16bit_maps(L, mode, geo, train):
    """CrossShapeAt forward + masked CE + backward with and without the 16-bit exchange, same weights, inputs and masks."""
    from csn_amd import tuning
    from csn_amd.csa_models import get_model
    rng = np.random.default_rng(31)
    B, K, n_cls = 2, 2, 7
    C = geo.get("d_model", 256)
    N = geo.get("block", 500) * geo.get("n_blocks", 20)
    torch.manual_seed(5)
    model = get_model("csa", n_cls, geo.get("n_head", 1), K, **{k: v for k, v in geo.items() if k != "n_head"}).cuda().train(train)
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
