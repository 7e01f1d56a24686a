"""The four-wave attention forward (csn_amd/csrc/attn_fwd_x4.hip: 32 queries per wave on v_mfma_f32_32x32x16_bf16, one wave per
SIMD, K / V by LDS-DMA) — measured slower than the eight-wave kernel and therefore off by default — against that kernel on the
same inputs (MID-FC/csa_models.py:138-144): Ctx, lse and the kept scores to the bf16x3 product error (the two kernels sum the
256 channels of a score in another order: ~1e-5 of the operands' scale, see gemm_bf16x3.hip), with dropout (the same counter-based
masks), several heads, slot maps and a row that ends inside the last block."""
import math

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


def _planes(kv, T, nb):
    S, R, _ = kv.shape
    x = torch.zeros((S, R, nb, 512), device=kv.device)
    for b in range(nb):
        n_b = min(T, kv.shape[2] - b * T)
        x[:, :, b, :n_b] = kv[:, :, b * T:b * T + n_b]
    x = x.view(S, R, nb, 16, 32)
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    return torch.stack((hi, lo), dim=4).reshape(S, R, nb * 1024).contiguous()


@pytest.mark.parametrize("S,E,H,T,nb,NP,drop,keep", [(3, 5, 1, 500, 2, 1000, 0.1, True), (2, 3, 2, 500, 2, 1000, 0.0, True),
                                                     (2, 4, 1, 500, 3, 1300, 0.1, True), (2, 2, 1, 100, 3, 300, 0.2, False),
                                                     (1, 1, 1, 36, 2, 72, 0.0, True)])
@pytest.mark.parametrize("form", [1, 2])          # 1: four waves x 32 queries (attn_fwd_x4.hip); 2: 4 query groups x 2 channel halves (attn_fwd_x8.hip)
def test_four_wave_forward_equals_the_eight_wave_forward(L, S, E, H, T, nb, NP, drop, keep, form):
    lib = L.lib()
    d, D, Tp = 256, 256 * H, (T + 31) // 32 * 32
    rng = np.random.default_rng(S * 100 + T)
    q = torch.from_numpy((rng.standard_normal((S, D, NP)) / 16).astype(np.float32)).cuda()
    kv = torch.from_numpy(rng.standard_normal((S, 2 * D, NP)).astype(np.float32)).cuda()
    kvt = _planes(kv, T, nb)
    ldp = nb * 1024
    qs = torch.from_numpy(rng.integers(0, S, size=E).astype(np.int32)).cuda()
    ks = torch.from_numpy(rng.integers(0, S, size=E).astype(np.int32)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for x4 in (0, form):
        lib.csn_dev_set(L.DEV_ATTN_X4, x4)
        try:
            ctx = torch.full((E, D, NP), float("nan"), device="cuda")
            lse = torch.full((E, H, nb * T), float("nan"), device="cuda")
            sc = torch.full((E, H, nb, T, Tp), float("nan"), device="cuda") if keep else None
            L.check(lib.csn_block_attn_fwd_f32(q.data_ptr(), kvt.data_ptr(), kvt.data_ptr() + 2 * D * ldp, D * NP, 2 * D * ldp,
                                               qs.data_ptr(), ks.data_ptr(), NP, ctx.data_ptr(), D * NP, None if sc is None else sc.data_ptr(),
                                               lse.data_ptr(), E, H, d, T, nb, Tp, 8.0, drop, 987654321, 1, ldp, st))
            torch.cuda.synchronize()
            outs.append((ctx.cpu(), lse.cpu(), None if sc is None else sc.cpu()))
        finally:
            lib.csn_dev_set(L.DEV_ATTN_X4, 0)
    (c0, l0, s0), (c1, l1, s1) = outs
    assert torch.equal(torch.isnan(c0), torch.isnan(c1)) and torch.equal(torch.isnan(l0), torch.isnan(l1))     # the same elements are written
    ok = ~torch.isnan(c0)
    assert torch.isfinite(c0[ok]).all()
    dc = (c0[ok] - c1[ok]).abs().max().item()
    assert dc <= 5e-5 * max(1.0, c0[ok].abs().max().item()), dc
    okl = ~torch.isnan(l0)
    dl = (l0[okl] - l1[okl]).abs().max().item()
    assert dl <= 5e-5 * max(1.0, l0[okl].abs().max().item()), dl
    if keep:
        assert torch.equal(torch.isnan(s0), torch.isnan(s1))
        oks = ~torch.isnan(s0) & torch.isfinite(s0)
        assert torch.equal(torch.isinf(s0), torch.isinf(s1))
        ds = (s0[oks] - s1[oks]).abs().max().item()
        assert ds <= 5e-5 * max(1.0, s0[oks].abs().max().item()), ds
