"""Stand-alone scaled-dot-product attention on (B, H, T, d) tensors through the HIP block-attention kernels
(ScaledDotProductAttention.forward, MID-FC/csa_models.py:138-144).  The hot path never goes through here —
MultiHeadAttention feeds the kernels channel-major data directly — this exists so the reference's public
class keeps working as a drop-in, and to materialise the probabilities when a caller asks for them."""
from __future__ import annotations

import torch

from . import _lib
from . import functional as CF


def _up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


class _SDPA(torch.autograd.Function):
    """softmax(q k^T / temperature) v for (B, H, Tq, d) x (B, H, Tk, d) x (B, H, Tk, d): every (batch, head) pair is one
    evaluation of the cross-length entry points (csn_cross_attn_fwd_f32 / _bwd_f32); Tq and Tk are arbitrary."""

    @staticmethod
    def forward(ctx, q, k, v, temperature: float, p_drop: float = 0.0):
        CF._need_cuda(q, k, v)
        B, H, Tq, d = q.shape
        Tk = k.shape[2]
        if v.shape[2] != Tk:
            raise ValueError("keys and values must have the same length")
        S = B * H
        Tq4, Tk4, Tp = _up(Tq, 4), _up(Tk, 4), _up(Tk, 32)

        def cm(x, T, Tpad, scale=1.0):     # channel-major maps [d][T], zero points appended up to a multiple of 4
            out = torch.zeros((S, d, Tpad), device=x.device, dtype=torch.float32)
            out[:, :, :T] = (x / scale if scale != 1.0 else x).reshape(S, T, d).transpose(1, 2)
            return out

        qm = cm(q, Tq, Tq4, temperature)   # the scale divides q before the product (csa_models.py:139)
        km, vm = cm(k, Tk, Tk4), cm(v, Tk, Tk4)
        att = torch.empty((S, d, Tq4), device=q.device, dtype=torch.float32)
        lse = torch.empty((S, 1, Tq4), device=q.device, dtype=torch.float32)
        scores = torch.empty((S, 1, Tq4, Tp), device=q.device, dtype=torch.float32)
        L = _lib.lib()
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p_drop > 0 else 0
        _lib.check(L.csn_cross_attn_fwd_f32(CF._ptr(qm), CF._ptr(km), CF._ptr(vm), d * Tq4, d * Tk4, Tq4, Tk4, CF._ptr(att),
                                            d * Tq4, CF._ptr(scores), CF._ptr(lse), S, 1, d, Tq4, Tk, Tp, CF.RESCALE_THRESHOLD,
                                            p_drop, seed, CF._stream()), "csn_cross_attn_fwd_f32")
        # P[q][key] = exp(S[q][key] - lse[q])   (the un-dropped probabilities; with dropout the reference returns the
        # dropped ones — every caller in the reference discards this tensor)
        prob = torch.exp(scores[:, 0, :Tq, :Tk] - lse[:, 0, :Tq, None]).reshape(B, H, Tq, Tk)
        ctx.save_for_backward(qm, km, vm, att, lse, scores)
        ctx.temperature = temperature
        ctx.drop = (p_drop, seed)
        ctx.dims = (B, H, Tq, Tk, d, Tp)
        ctx.mark_non_differentiable(prob)
        return att[:, :, :Tq].transpose(1, 2).reshape(B, H, Tq, d), prob

    @staticmethod
    def backward(ctx, dout, _dprob):
        qm, km, vm, att, lse, scores = ctx.saved_tensors
        B, H, Tq, Tk, d, Tp = ctx.dims
        S = B * H
        Tq4, Tk4 = qm.shape[2], km.shape[2]
        datt = torch.zeros((S, d, Tq4), device=dout.device, dtype=torch.float32)
        datt[:, :, :Tq] = dout.reshape(S, Tq, d).transpose(1, 2)
        dq = torch.empty((S, d, Tq4), device=dout.device, dtype=torch.float32)
        dk, dv = (torch.empty((S, d, Tk4), device=dout.device, dtype=torch.float32) for _ in range(2))
        work = scores.clone()                      # backward overwrites the scores with the (dropped) probabilities
        dscores = torch.empty_like(scores)
        delta = torch.empty((S, 1, Tq4), device=dout.device, dtype=torch.float32)
        L = _lib.lib()
        _lib.check(L.csn_cross_attn_bwd_f32(CF._ptr(datt), CF._ptr(att), d * Tq4, CF._ptr(qm), CF._ptr(km), CF._ptr(vm), d * Tq4,
                                            d * Tk4, Tq4, Tk4, CF._ptr(work), CF._ptr(dscores), CF._ptr(lse), CF._ptr(delta),
                                            CF._ptr(dq), CF._ptr(dk), CF._ptr(dv), d * Tq4, d * Tk4, S, 1, d, Tq4, Tk, Tp,
                                            ctx.drop[0], ctx.drop[1], CF._stream()), "csn_cross_attn_bwd_f32")
        back = lambda g, T: g[:, :, :T].transpose(1, 2).reshape(B, H, T, d)
        return back(dq, Tq) / ctx.temperature, back(dk, Tk), back(dv, Tk), None, None


def sdpa_block(q, k, v, temperature: float, p_drop: float = 0.0):
    return _SDPA.apply(q, k, v, temperature, float(p_drop))


sdpa_cross = sdpa_block        # same entry: query and key counts may differ (MinkowskiNet/models/attention.py:59-73)


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
