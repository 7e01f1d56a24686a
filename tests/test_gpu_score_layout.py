"""Tile-major score storage (csn_set_thread_score_layout, include/csn_hip.h): the forward's scores and the P / dS planes of the
backward stored per block as [key tile][query][32 keys] instead of [query][key] rows — the same arithmetic in another place, so
a training step must come out bit for bit the same (MID-FC/csa_models.py:138-144 forward, its autograd backward)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from csn_amd import _lib
    _lib.build()
    _lib.check(_lib.lib().csn_set_math_mode(1))
    return _lib


def test_capability_bit_and_flag_validation(L):
    lib = L.lib()
    assert lib.csn_attn_bwd_grouping(256, 500) & 16              # d = 256: the dV / dK products run on the 256 x 256 tiles
    assert not (lib.csn_attn_bwd_grouping(96, 500) & 16)         # narrow heads: 128 x 128 tiles, row-major planes
    assert lib.csn_set_thread_score_layout(2) == -1
    assert lib.csn_get_thread_score_layout() == 0
    L.check(lib.csn_set_math_mode(0))
    try:
        assert not (lib.csn_attn_bwd_grouping(256, 500) & 16)    # exact fp32 mode: no planes at all
    finally:
        L.check(lib.csn_set_math_mode(1))


@pytest.mark.parametrize("N,nb,train", [(1500, 3, True), (1300, 3, True), (1000, 2, False)])
def test_training_step_is_bitwise_the_same_in_both_layouts(L, N, nb, train):
    from csn_amd import tuning
    from csn_amd.csa_models import get_model
    from oracle import csa_oracle as orc
    rng = np.random.default_rng(71)
    B, K, n_cls, C = 2, 2, 7, 256
    torch.manual_seed(3)
    model = get_model("csa", n_cls, 1, K, block=500, n_blocks=nb if N == nb * 500 else None).cuda()
    model = model.train() if train else model.eval()
    nbf = torch.from_numpy(rng.standard_normal((B, K + 1, C, N, 1)).astype(np.float32)).cuda()
    x = nbf[:, 0].contiguous()
    lab = torch.from_numpy(rng.integers(0, n_cls, size=(B, N))).cuda()
    outs = []
    for tm in (False, True):
        with tuning.override(tile_major_scores=tm):
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
