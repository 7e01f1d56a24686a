"""dV / dK on the P / dS tile planes as output-stationary streams (csn_amd/csrc/dkv_stream.hip behind csn_block_attn_bwd_dkv_f32:
the backward of MID-FC/csa_models.py:140-142 w.r.t. v and k, bf16x3 mode, d_head = 256) against the grouped 256 x 256-tile GEMM
route it replaces (development switch CSN_DEV_DKV_STREAM = 0).  Both form every sum in the same order, so the results must be
the same BITS — at the entry point (grouped, ungrouped, accumulating, ragged last block, two heads, short blocks) and over a
training step of the module.  The GEMM route itself is held to float64 in tests/test_gpu_kernels.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from csn_amd import _lib
    _lib.build()
    _lib.check(_lib.lib().csn_set_math_mode(1))
    yield _lib
    _lib.lib().csn_dev_set(_lib.DEV_DKV_STREAM, 1)


def _planes(gen, shape, n_keys):
    """random [.., query][key tile][hi 32 | lo 32] planes stored in an fp32 buffer (two bf16 per float), like the dQ kernel leaves
    them: keys beyond the block end are zeros"""
    v = torch.randn(shape, device="cuda", generator=gen) * 0.05
    v[..., n_keys:] = 0
    hi = v.bfloat16()
    lo = (v - hi.float()).bfloat16()
    t = torch.stack((hi.view(*shape[:-1], shape[-1] // 32, 32), lo.view(*shape[:-1], shape[-1] // 32, 32)), dim=-2)
    return t.reshape(*shape[:-1], 2 * shape[-1]).contiguous().view(torch.float32)       # [..][Tp floats] = Tp * 2 bf16


@pytest.mark.parametrize("T,nb,ld,H,grouped,accumulate", [
    (500, 3, 1500, 1, True, 0),        # the reference's blocks, grouped by key / value slot (the training step's call)
    (500, 3, 1300, 1, True, 0),        # the row ends inside the last block (300 points)
    (500, 2, 1000, 2, False, 1),       # two heads, one launch item per evaluation, accumulating into the slot maps
    (256, 2, 512, 1, True, 0),         # one key half
    (64, 3, 192, 1, False, 0),         # short blocks, score pitch 256
])
def test_entry_point_gives_the_gemm_routes_bits(L, T, nb, ld, H, grouped, accumulate):
    from csn_amd import functional as CF
    lib = L.lib()
    d, S, E = 256, 3, 5
    D = H * d
    Tp = max(256, (T + 31) // 32 * 32)
    g = torch.Generator(device="cuda").manual_seed(T + nb + H)
    dctx = torch.randn((E, D, ld), device="cuda", generator=g)
    q = torch.randn((S, D, ld), device="cuda", generator=g) * 0.25
    probs, dsc = _planes(g, (E, H, nb, T, Tp), T), _planes(g, (E, H, nb, T, Tp), T)
    q_index = torch.tensor([0, 1, 2, 0, 1], device="cuda", dtype=torch.int32)
    kv_slot = torch.tensor([0, 0, 1, 2, 2], device="cuda", dtype=torch.int32)       # evaluations 0,1 / 2 / 3,4 share a slot
    items = torch.arange(E, device="cuda", dtype=torch.int32)
    grp_off = torch.tensor([0, 2, 3, 5], device="cuda", dtype=torch.int32)
    outs = []
    for stream in (0, 1):
        lib.csn_dev_set(L.DEV_DKV_STREAM, stream)
        dk = torch.full((S, D, ld), 0.5, device="cuda")
        dv = torch.full((S, D, ld), -0.25, device="cuda")
        if grouped:
            rc = lib.csn_block_attn_bwd_dkv_f32(CF._ptr(dctx), D * ld, CF._ptr(q), D * ld, CF._ptr(q_index), ld, CF._ptr(probs), CF._ptr(dsc),
                                                CF._ptr(dk), CF._ptr(dv), D * ld, CF._ptr(kv_slot), CF._ptr(kv_slot), accumulate, CF._ptr(items), E, H, d,
                                                T, nb, Tp, 0, 0, 0, 0, 1, CF._ptr(grp_off), 3, CF._stream())
        else:
            # ungrouped launches must not repeat a slot: one launch per "colour"
            rc = 0
            for sel in ([0, 2, 3], [1, 4]):
                ids = torch.tensor(sel, device="cuda", dtype=torch.int32)
                rc |= lib.csn_block_attn_bwd_dkv_f32(CF._ptr(dctx), D * ld, CF._ptr(q), D * ld, CF._ptr(q_index), ld, CF._ptr(probs), CF._ptr(dsc),
                                                     CF._ptr(dk), CF._ptr(dv), D * ld, CF._ptr(kv_slot), CF._ptr(kv_slot), accumulate, CF._ptr(ids), len(sel),
                                                     H, d, T, nb, Tp, 0, 0, 0, 0, 1, None, 0, CF._stream())
        L.check(rc, "csn_block_attn_bwd_dkv_f32")
        torch.cuda.synchronize()
        outs.append((dk.clone(), dv.clone()))
    lib.csn_dev_set(L.DEV_DKV_STREAM, 1)
    assert torch.isfinite(outs[0][0]).all() and outs[0][0].abs().max() > 0
    n_valid = min(ld, nb * T)
    if not accumulate and grouped:
        assert not torch.equal(outs[0][0][:, :, :n_valid], torch.full_like(outs[0][0][:, :, :n_valid], 0.5))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("N,nb,train,heads", [(1500, 3, True, 1), (1300, 3, True, 1), (1000, 2, False, 2)])
def test_training_step_is_bitwise_the_same_on_either_route(L, N, nb, train, heads):
    from csn_amd.csa_models import get_model
    from oracle import csa_oracle as orc
    lib = L.lib()
    rng = np.random.default_rng(72)
    B, K, n_cls, C = 2, 2, 7, 256
    torch.manual_seed(3)
    model = get_model("csa", n_cls, heads, K, block=500, n_blocks=nb if N == nb * 500 else None).cuda()
    model = model.train() if train else model.eval()
    nbf = torch.from_numpy(rng.standard_normal((B, K + 1, C, N, 1)).astype(np.float32)).cuda()
    x = nbf[:, 0].contiguous()
    lab = torch.from_numpy(rng.integers(0, n_cls, size=(B, N))).cuda()
    outs = []
    for stream in (0, 1):
        lib.csn_dev_set(L.DEV_DKV_STREAM, stream)
        for prm in model.parameters():
            prm.grad = None
        torch.manual_seed(5)
        logits = model(x, "train", nbf)
        orc.masked_ce_loss(logits, lab).backward()
        outs.append((logits.detach().clone(), [p.grad.clone() for p in model.parameters() if p.grad is not None]))
    lib.csn_dev_set(L.DEV_DKV_STREAM, 1)
    assert torch.isfinite(outs[0][0]).all() and len(outs[0][1]) == 11
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)
