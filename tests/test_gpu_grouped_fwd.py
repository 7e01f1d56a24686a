"""The attention forward with its evaluations grouped by query slot (csn_block_attn_fwd_grouped_f32: the pre-scaled queries of a
128-query tile staged once, the group's evaluations run one after the other with the operand in registers) against the
ungrouped call: the same arithmetic in the same order, so every output — Ctx, lse, the kept scores — and a whole training step
must be the same BITS (MID-FC/csa_models.py:138-144; the query shape of a CSA step serves K+2 evaluations, :210, :232-237)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[1, 2], ids=["bf16x3", "bf16"])
def L(request):
    from csn_amd import _lib
    _lib.build()
    _lib.check(_lib.lib().csn_set_math_mode(request.param))
    yield _lib
    _lib.lib().csn_set_math_mode(1)


@pytest.mark.parametrize("T,nb,ld,H,d,drop", [(500, 3, 1500, 1, 256, 0.1), (500, 3, 1300, 1, 256, 0.0), (100, 4, 400, 2, 64, 0.1),
                                               (500, 2, 1000, 1, 96, 0.1)])
def test_entry_point_gives_the_ungrouped_calls_bits(L, T, nb, ld, H, d, drop):
    from csn_amd import functional as CF
    lib = L.lib()
    mode = lib.csn_get_math_mode()
    npl = 2 if mode == 1 else 1
    S, D = 4, H * d
    Tp = (T + 31) // 32 * 32
    g = torch.Generator(device="cuda").manual_seed(7 + T + d)
    x = torch.randn((S, 256, ld), device="cuda", generator=g)
    w = torch.randn((3 * D, 256), device="cuda", generator=g) / 16
    ldp = nb * 512 * npl
    qs = torch.empty((S, D, ld), device="cuda")
    kv = torch.zeros((S, 2 * D, ldp), device="cuda", dtype=torch.bfloat16)
    L.check(lib.csn_project_f32(CF._ptr(x), 256 * ld, ld, CF._ptr(w), D, 256, CF._ptr(qs), D * ld, ld, S, ld, D, float(d) ** 0.5, 0, 0, CF._stream()))
    L.check(lib.csn_project_f32(CF._ptr(x), 256 * ld, ld, CF._ptr(w[D:].contiguous()), 2 * D, 256, CF._ptr(kv), 2 * D * ldp, ldp, S, ld, 0, 1.0, 2, T,
                                CF._stream()))
    # 7 evaluations over 4 slots: slot 0 queries four of them, slot 2 two, slot 3 one
    q_idx = torch.tensor([0, 2, 0, 3, 0, 2, 0], device="cuda", dtype=torch.int32)
    k_idx = torch.tensor([0, 2, 1, 3, 2, 0, 3], device="cuda", dtype=torch.int32)
    items = torch.tensor([0, 2, 4, 6, 1, 5, 3], device="cuda", dtype=torch.int32)          # biggest group first
    off = torch.tensor([0, 4, 6, 7], device="cuda", dtype=torch.int32)
    E = 7
    outs = []
    for grouped in (False, True):
        att = torch.full((E, D, ld), 7.0, device="cuda")
        lse = torch.full((E, H, nb * T), 7.0, device="cuda")
        sc = torch.full((E, H, nb, T, Tp), 7.0, device="cuda")
        args = (CF._ptr(qs), CF._ptr(kv), kv.data_ptr() + 2 * D * ldp, D * ld, 2 * D * ldp, CF._ptr(q_idx), CF._ptr(k_idx), ld, CF._ptr(att), D * ld,
                CF._ptr(sc), CF._ptr(lse), E, H, d, T, nb, Tp, 8.0, drop, 991, 1, ldp)
        if grouped:
            L.check(lib.csn_block_attn_fwd_grouped_f32(*args, CF._ptr(items), CF._ptr(off), 3, CF._stream()), "grouped")
        else:
            L.check(lib.csn_block_attn_fwd_f32(*args, CF._stream()), "ungrouped")
        torch.cuda.synchronize()
        outs.append((att, lse, sc))
    assert torch.isfinite(outs[0][0]).all()
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("N,nb,train", [(1500, 3, True), (1300, 3, True), (1000, 2, False)])
def test_training_step_is_bitwise_the_same_grouped_and_ungrouped(L, N, nb, train):
    from csn_amd import tuning
    from csn_amd.csa_models import get_model
    from oracle import csa_oracle as orc
    rng = np.random.default_rng(73)
    B, K, n_cls, C = 2, 2, 7, 256
    torch.manual_seed(3)
    model = get_model("csa", n_cls, 1, K, block=500, n_blocks=nb if N == nb * 500 else None).cuda()
    model = model.train() if train else model.eval()
    nbf = torch.from_numpy(rng.standard_normal((B, K + 1, C, N, 1)).astype(np.float32)).cuda()
    x = nbf[:, 0].contiguous()
    lab = torch.from_numpy(rng.integers(0, n_cls, size=(B, N))).cuda()
    outs = []
    for grouped in (False, True):
        with tuning.override(grouped_fwd=grouped):
            for prm in model.parameters():
                prm.grad = None
            torch.manual_seed(5)
            logits = model(x, "train", nbf)
            orc.masked_ce_loss(logits, lab).backward()
            outs.append((logits.detach().clone(), [p.grad.clone() for p in model.parameters() if p.grad is not None]))
    assert torch.isfinite(outs[0][0]).all() and len(outs[0][1]) == 11
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)
