"""Stand-alone scaled-dot-product attention on (B, H, T, d) tensors through the HIP block-attention kernels
(ScaledDotProductAttention.forward, MID-FC/csa_models.py:138-144).  The hot path never goes through here —
MultiHeadAttention feeds the kernels channel-major data directly — this exists so the reference's public
class keeps working as a drop-in, and to materialise the probabilities when a caller asks for them."""
from __future__ import annotations

import torch

from . import _lib
from . import functional as CF


class _SDPA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, temperature: float, p_drop: float = 0.0):
        CF._need_cuda(q, k, v)
        B, H, T, d = q.shape
        Tk = k.shape[2]
        if T != Tk:
            raise NotImplementedError("query and key blocks must have the same length")
        S = B * H
        # channel-major maps [d][T]; the scale divides q before the product (csa_models.py:139)
        qm = (q / temperature).reshape(S, T, d).transpose(1, 2).contiguous()
        km = k.reshape(S, T, d).transpose(1, 2).contiguous()
        vm = v.reshape(S, T, d).transpose(1, 2).contiguous()
        Tp = (T + 31) // 32 * 32
        att = torch.empty((S, d, T), device=q.device, dtype=torch.float32)
        lse = torch.empty((S, 1, T), device=q.device, dtype=torch.float32)
        scores = torch.empty((S, 1, 1, T, Tp), device=q.device, dtype=torch.float32)
        L = _lib.lib()
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p_drop > 0 else 0
        _lib.check(L.csn_block_attn_fwd_f32(CF._ptr(qm), CF._ptr(km), CF._ptr(vm), d * T, d * T, None, None, T,
                                            CF._ptr(att), d * T, CF._ptr(scores), CF._ptr(lse), S, 1, d, T, 1, Tp,
                                            CF.RESCALE_THRESHOLD, p_drop, seed, 0, 0, CF._stream()), "csn_block_attn_fwd_f32")
        # P[q][key] = exp(S[q][key] - lse[q])   (the un-dropped probabilities; with dropout the reference returns the
        # dropped ones — every caller in the reference discards this tensor)
        prob = torch.exp(scores[:, 0, 0, :, :T] - lse[:, 0, :, None]).reshape(B, H, T, T)
        ctx.save_for_backward(qm, km, vm, att, lse, scores)
        ctx.temperature = temperature
        ctx.drop = (p_drop, seed)
        ctx.dims = (B, H, T, d, Tp)
        ctx.mark_non_differentiable(prob)
        return att.transpose(1, 2).reshape(B, H, T, d), prob

    @staticmethod
    def backward(ctx, dout, _dprob):
        qm, km, vm, att, lse, scores = ctx.saved_tensors
        B, H, T, d, Tp = ctx.dims
        S = B * H
        datt = dout.reshape(S, T, d).transpose(1, 2).contiguous()
        dq, dk, dv = (torch.empty((S, d, T), device=dout.device, dtype=torch.float32) for _ in range(3))
        work = scores.clone()                      # backward overwrites the scores with the (dropped) probabilities
        dscores = torch.empty_like(scores)
        delta = torch.empty((S, 1, T), device=dout.device, dtype=torch.float32)
        L = _lib.lib()
        _lib.check(L.csn_block_attn_bwd_dq_f32(CF._ptr(datt), CF._ptr(att), d * T, CF._ptr(km), CF._ptr(vm), d * T, None, T,
                                               CF._ptr(work), CF._ptr(dscores), CF._ptr(lse), CF._ptr(delta), CF._ptr(dq),
                                               d * T, None, 0, None, S, 1, d, T, 1, Tp, ctx.drop[0], ctx.drop[1], 0, 0, 0, 0, 0,
                                               CF._stream()), "csn_block_attn_bwd_dq_f32")
        _lib.check(L.csn_block_attn_bwd_dkv_f32(CF._ptr(datt), d * T, CF._ptr(qm), d * T, None, T, CF._ptr(work),
                                                CF._ptr(dscores), CF._ptr(dk), CF._ptr(dv), d * T, None, None, 0, None, S, 1,
                                                d, T, 1, Tp, 0, 0, 0, 0, 0, CF._stream()), "csn_block_attn_bwd_dkv_f32")
        back = lambda g: g.transpose(1, 2).reshape(B, H, T, d)
        return back(dq) / ctx.temperature, back(dk), back(dv), None, None


def sdpa_block(q, k, v, temperature: float, p_drop: float = 0.0):
    return _SDPA.apply(q, k, v, temperature, float(p_drop))


def last_block_probabilities(mha, Q, K):
    """Probabilities of the last block, (B, H, T, T), as the reference returns them (csa_models.py:125)."""
    geo = mha.geometry()
    lo, hi = (geo.n_blocks - 1) * geo.block, geo.n_blocks * geo.block

    def chunk(x):
        if x.dim() == 4:
            x = x.squeeze(-1)
        return x[:, :, lo:hi].to(CF_device(), torch.float32).permute(0, 2, 1)      # (B, T, C)

    with torch.no_grad():
        B = Q.shape[0]
        q = mha.w_qs(chunk(Q)).view(B, geo.block, geo.n_head, geo.d_head).transpose(1, 2)
        k = mha.w_ks(chunk(K)).view(B, geo.block, geo.n_head, geo.d_head).transpose(1, 2)
        return sdpa_block(q.contiguous(), k.contiguous(), k.contiguous(), float(geo.d_head) ** 0.5)[1]


def CF_device():
    return torch.device("cuda")
