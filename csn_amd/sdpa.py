"""Stand-alone scaled-dot-product attention on (B, H, T, d) tensors through the HIP block-attention kernels
(ScaledDotProductAttention.forward, MID-FC/csa_models.py:138-144).  The hot path never goes through here —
MultiHeadAttention feeds the kernels channel-major data directly — this exists so the reference's public
class keeps working as a drop-in, and to materialise the probabilities when a caller asks for them."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib
from . import functional as CF


def _up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


_M32 = 0xFFFFFFFF


def _mix32(h: torch.Tensor) -> torch.Tensor:
    """csn_mix32 (csrc/csn_common.h) on int64 tensors holding 32-bit values."""
    h = h ^ (h >> 16)
    h = (h * 0x85EBCA6B) & _M32
    h = h ^ (h >> 13)
    h = (h * 0xC2B2AE35) & _M32
    return h ^ (h >> 16)


def attention_keep_mask(n_evals: int, n_queries: int, n_keys: int, score_pitch: int, seed: int, p: float, device) -> torch.Tensor:
    """The keep mask the attention kernels apply to the probabilities of ``n_evals`` single-block evaluations
    (csn_block_salt / csn_pair_hash of csrc/csn_common.h, restated with torch integer ops on the device): bool
    (n_evals, n_queries, n_keys).  Only used to hand a caller the DROPPED probabilities the reference returns in train mode
    (csa_models.py:141-144); the kernels regenerate the mask themselves."""
    e = torch.arange(n_evals, device=device, dtype=torch.int64)
    s0, s1 = seed & _M32, (seed >> 32) & _M32
    salt = _mix32((_mix32((e & _M32) ^ s0) + ((e >> 32) ^ s1) + 0x9E3779B9) & _M32).view(n_evals, 1, 1)
    q = torch.arange(n_queries, device=device, dtype=torch.int64).view(1, n_queries, 1)
    key = torch.arange(n_keys, device=device, dtype=torch.int64).view(1, 1, n_keys)
    pair = ((key >> 1) * max(score_pitch, n_queries) + q) & _M32
    h = _mix32(pair ^ salt)
    field = torch.where((key & 1) == 1, h >> 16, h & 0xFFFF)
    # (the threshold is formed in fp32 like csn_drop_threshold16)
    return field >= int(torch.tensor(p, dtype=torch.float32).mul(65536.0).item())


class _SDPA(torch.autograd.Function):
    """softmax(q k^T / temperature) v for (B, H, Tq, d) x (B, H, Tk, d) x (B, H, Tk, d): every (batch, head) pair is one
    evaluation of the cross-length entry points (csn_cross_attn_fwd_f32 / _bwd_f32); Tq and Tk are arbitrary."""

    @staticmethod
    def forward(ctx, q, k, v, temperature: float, p_drop: float = 0.0):
        CF._need_cuda(q, k, v)
        ctx.mode = CF.current_mode()               # the backward runs on autograd's threads: it re-opens this mode there
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
        seed = CF.draw_seeds(1)[0] if p_drop > 0 else 0
        _lib.check(L.csn_cross_attn_fwd_f32(CF._ptr(qm), CF._ptr(km), CF._ptr(vm), d * Tq4, d * Tk4, Tq4, Tk4, CF._ptr(att),
                                            d * Tq4, CF._ptr(scores), CF._ptr(lse), S, 1, d, Tq4, Tk, Tp, CF.RESCALE_THRESHOLD,
                                            p_drop, seed, CF._stream()), "csn_cross_attn_fwd_f32")
        # P[q][key] = exp(S[q][key] - lse[q]); in train mode the reference returns the DROPPED probabilities
        # (attn = dropout(softmax(..)), csa_models.py:141-144): the kernel's mask is rebuilt here, outside the hot path
        prob = torch.exp(scores[:, 0, :Tq, :Tk] - lse[:, 0, :Tq, None])
        if p_drop > 0:
            prob = prob * attention_keep_mask(S, Tq, Tk, Tp, seed, p_drop, q.device) / (1.0 - p_drop)
        prob = prob.reshape(B, H, Tq, Tk)
        ctx.save_for_backward(qm, km, vm, att, lse, scores)
        ctx.temperature = temperature
        ctx.drop = (p_drop, seed)
        ctx.dims = (B, H, Tq, Tk, d, Tp)
        ctx.mark_non_differentiable(prob)
        return att[:, :, :Tq].transpose(1, 2).reshape(B, H, Tq, d), prob

    @staticmethod
    def backward(ctx, dout, _dprob):
        with CF.math_mode(CF.backward_mode(ctx.mode)):
            return _SDPA._backward(ctx, dout, _dprob)

    @staticmethod
    def _backward(ctx, dout, _dprob):
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
    """(out, probabilities).  q / k of width d_k and v of width d_v run at one kernel head width with zero channels appended
    (scores unchanged; the extra output channels are zero and cut off again)."""
    dk, dv = q.shape[-1], v.shape[-1]
    d = CF.kernel_head_width(max(dk, dv))
    if dk != d:
        q, k = F.pad(q, (0, d - dk)), F.pad(k, (0, d - dk))
    if dv != d:
        v = F.pad(v, (0, d - dv))
    out, prob = _SDPA.apply(q, k, v, temperature, float(p_drop))
    return (out if dv == d else out[..., :dv]), prob


sdpa_cross = sdpa_block        # same entry: query and key counts may differ (MinkowskiNet/models/attention.py:59-73)


def last_block_probabilities(mha, Q, K):
    """Probabilities of the last block, (B, H, T, T), as the reference returns them (csa_models.py:125): the last block is
    projected again (HIP GEMM, csn_project_f32) and scored by the stand-alone entry.  In train mode these are dropped
    probabilities under a mask of their own draw (the hot path does not keep its masks)."""
    geo = mha.geometry(n_points=Q.shape[2])
    lo, hi = (geo.n_blocks - 1) * geo.block, geo.n_points            # (a ragged last block ends with the row)

    def chunk(x):
        if x.dim() == 4:
            x = x.squeeze(-1)
        return x[:, :, lo:hi].to(CF_device(), torch.float32).contiguous()         # (B, C, T) channel-major

    with torch.no_grad():
        B = Q.shape[0]
        heads = lambda m: m.view(B, geo.n_head, mha.d_k, hi - lo).transpose(2, 3).contiguous()        # (B, H, T, d_k)
        q = heads(CF.project(chunk(Q), mha.w_qs.weight.contiguous()))
        k = heads(CF.project(chunk(K), mha.w_ks.weight.contiguous()))
        p_drop = mha.attention.dropout.p if mha.training else 0.0
        return sdpa_block(q, k, k, float(mha.d_k) ** 0.5, p_drop)[1]


def CF_device():
    return torch.device("cuda")
