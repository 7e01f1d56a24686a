"""Drop-in for ``MinkowskiNet/models/attention.py`` (the MinkowskiNet variant of the cross-shape attention layer) on the
MI355X kernels: unchunked multi-head attention between two point sets of DIFFERENT sizes, point-major inputs, gradients
flowing to queries, keys and values (the features come from a trainable backbone there).

Reference (all under /root/reference):
  * ``MultiHeadAttention``            MinkowskiNet/models/attention.py:9-56   (used per shape pair by hrnet.py:378-410, 456-470)
  * ``ScaledDotProductAttention``     MinkowskiNet/models/attention.py:59-73
  * ``ScaledDotProduct``              MinkowskiNet/models/attention.py:76-104  (two 256-vectors: stays a torch op)

Same class names, constructor arguments, parameter names and ``forward(q, k, v) -> (out, attn)`` as the reference.  The
arithmetic runs in libcsn_hip.so (``csn_cross_attn_fwd_f32`` / ``csn_cross_attn_bwd_f32`` + the projection / out-projection
kernels of the MID-FC path); there is no eager fallback.  SURVEY.md §8(f) rank 2.  The sparse-tensor branches of the
reference (MinkowskiEngine, absent here) are out of scope: callers pass the dense feature matrices (``features_at``).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from . import functional as CF


def _up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def _to_cm(x: torch.Tensor, n_pad: int) -> torch.Tensor:
    """(b, n, C) point-major -> (b, C, n_pad) channel-major fp32, zero points appended."""
    b, n, c = x.shape
    out = torch.zeros((b, c, n_pad), device=x.device, dtype=torch.float32)
    out[:, :, :n] = x.transpose(1, 2)
    return out


class _CrossMHA(torch.autograd.Function):
    """xhat = LayerNorm_noaffine(dropout(fc(Attn(Wq xq / sqrt(d), Wk xk, Wv xv))) + xq), channel-major, padded to % 4."""

    @staticmethod
    def forward(ctx, xq, xk, xv, w_qs, w_ks, w_vs, w_fc, n_head: int, d_head: int, p_attn: float, p_fc: float,
                want_attn: bool, keep: bool, q_lens=None, k_lens=None, temperature: float = 0.0):
        """q_lens / k_lens (int32 device tensors, one entry per batch item; q_lens % 4 == 0): a RAGGED batch — item i uses
        the first q_lens[i] query points and k_lens[i] key points of its zero-padded rows, in one launch chain
        (csn_varlen_attn_fwd_f32 / _bwd_f32)."""
        CF._need_cuda(xq, xk, xv, w_qs, w_ks, w_vs, w_fc)
        L = _lib.lib()
        ctx.mode = CF.current_mode()                 # the backward runs on autograd's threads: it re-opens this mode there
        b, lq, C = xq.shape
        lk = xk.shape[1]
        if xv.shape[1] != lk:
            raise ValueError("keys and values must have the same length")
        varlen = q_lens is not None
        H, d, D = n_head, d_head, n_head * d_head
        lq4, lk4 = _up(lq, 4), _up(lk, 4)
        Tp = _up(lk, 32)
        dev = xq.device
        temperature = temperature or float(d) ** 0.5     # (given when d_head is a kernel width that padded weights fill)
        ctx.temperature = temperature
        xq_cm, xk_cm = _to_cm(xq, lq4), _to_cm(xk, lk4)
        xv_cm = xk_cm if xv is xk else _to_cm(xv, lk4)
        q = CF.project(xq_cm, w_qs.contiguous(), div_rows=D, temperature=temperature)     # (b, D, lq4), pre-scaled
        k = CF.project(xk_cm, w_ks.contiguous())
        v = CF.project(xv_cm, w_vs.contiguous())
        seed_attn, seed_fc = CF.draw_seeds(2) if (p_attn > 0 or p_fc > 0) else (0, 0)
        # ragged batch: the context of padding queries is never written — it must read as zero downstream (fc + LayerNorm of
        # a zero row is a finite zero row, and nothing flows back from it)
        att = (torch.zeros if varlen else torch.empty)((b, D, lq4), device=dev, dtype=torch.float32)
        lse = torch.empty((b, H, lq4), device=dev, dtype=torch.float32)
        scores = torch.empty((b, H, lq4, Tp), device=dev, dtype=torch.float32) if (keep or want_attn) else None
        if varlen:
            _lib.check(L.csn_varlen_attn_fwd_f32(CF._ptr(q), CF._ptr(k), CF._ptr(v), D * lq4, D * lk4, lq4, lk4, CF._ptr(att),
                                                 D * lq4, CF._ptr(scores), CF._ptr(lse), b, H, d, lq4, lk, CF._ptr(q_lens),
                                                 CF._ptr(k_lens), Tp, CF.RESCALE_THRESHOLD, p_attn, seed_attn, CF._stream()),
                       "csn_varlen_attn_fwd_f32")
        else:
            _lib.check(L.csn_cross_attn_fwd_f32(CF._ptr(q), CF._ptr(k), CF._ptr(v), D * lq4, D * lk4, lq4, lk4, CF._ptr(att),
                                                D * lq4, CF._ptr(scores), CF._ptr(lse), b, H, d, lq4, lk, Tp,
                                                CF.RESCALE_THRESHOLD, p_attn, seed_attn, CF._stream()), "csn_cross_attn_fwd_f32")
        attn = None
        if want_attn:
            # the un-dropped probabilities (the reference returns the dropped ones in train mode; hrnet.py discards them)
            attn = torch.exp(scores[:, :, :lq, :lk] - lse[:, :, :lq, None])
        xhat = torch.empty((b, C, lq4), device=dev, dtype=torch.float32)
        rstd = torch.empty((b, lq4), device=dev, dtype=torch.float32)
        w_fc = w_fc.contiguous()
        _lib.check(L.csn_outproj_ln_fwd_f32(CF._ptr(att), D * lq4, CF._ptr(w_fc), CF._ptr(xq_cm), C * lq4, None, CF._ptr(xhat),
                                            C * lq4, CF._ptr(rstd), b, C, D, lq4, lq4, CF.LN_EPS, p_fc, seed_fc, None, None, 0, CF._stream()),
                   "csn_outproj_ln_fwd_f32")
        if keep:
            ctx.save_for_backward(xq_cm, xk_cm, xv_cm, w_qs, w_ks, w_vs, w_fc, q, k, v, att, lse, scores, xhat, rstd)
            ctx.dims = (b, lq, lk, C, H, d, Tp)
            ctx.drop = (p_attn, seed_attn, p_fc, seed_fc)
            ctx.lens = (q_lens, k_lens)
        if attn is not None:
            ctx.mark_non_differentiable(attn)
        return xhat, attn

    @staticmethod
    def backward(ctx, dxhat, _dattn):
        with CF.math_mode(CF.backward_mode(ctx.mode)):
            return _CrossMHA._backward(ctx, dxhat, _dattn)

    @staticmethod
    def _backward(ctx, dxhat, _dattn):
        xq_cm, xk_cm, xv_cm, w_qs, w_ks, w_vs, w_fc, q, k, v, att, lse, scores, xhat, rstd = ctx.saved_tensors
        b, lq, lk, C, H, d, Tp = ctx.dims
        p_attn, seed_attn, p_fc, seed_fc = ctx.drop
        L = _lib.lib()
        D = H * d
        lq4, lk4 = xq_cm.shape[2], xk_cm.shape[2]
        dev = xq_cm.device
        temperature = ctx.temperature
        dxhat = dxhat.contiguous()
        # LayerNorm + fc backward
        dz = torch.empty((b, C, lq4), device=dev, dtype=torch.float32)
        dz_res = torch.empty((b, C, lq4), device=dev, dtype=torch.float32) if p_fc > 0 else None
        datt = torch.empty((b, D, lq4), device=dev, dtype=torch.float32)
        dw_fc = torch.empty((C, D), device=dev, dtype=torch.float32)
        ws_n = L.csn_wgrad_workspace_floats(C, D, b, lq4)
        ws = torch.empty((ws_n,), device=dev, dtype=torch.float32)
        _lib.check(L.csn_outproj_ln_bwd_f32(CF._ptr(dxhat), CF._ptr(xhat), CF._ptr(rstd), C * lq4, CF._ptr(att), D * lq4,
                                            CF._ptr(w_fc.t().contiguous()), CF._ptr(dz), CF._ptr(dz_res), CF._ptr(datt),
                                            CF._ptr(dw_fc), CF._ptr(ws), ws_n, b, C, D, lq4, lq4, 0, p_fc, seed_fc, 0, 0, None,
                                            b, None, 1, CF._stream()), "csn_outproj_ln_bwd_f32")
        # attention backward: gradients to the projected queries, keys and values
        dscores = torch.empty_like(scores)
        delta = torch.empty((b, H, lq4), device=dev, dtype=torch.float32)
        q_lens, k_lens = ctx.lens
        varlen = q_lens is not None
        alloc = torch.zeros if varlen else torch.empty               # ragged batch: rows / columns of the padding stay zero
        dq = alloc((b, D, lq4), device=dev, dtype=torch.float32)
        dk = alloc((b, D, lk4), device=dev, dtype=torch.float32)
        dv = alloc((b, D, lk4), device=dev, dtype=torch.float32)
        work = scores.clone()                                        # the backward turns the scores into probabilities in place
        if varlen:
            _lib.check(L.csn_varlen_attn_bwd_f32(CF._ptr(datt), CF._ptr(att), D * lq4, CF._ptr(q), CF._ptr(k), CF._ptr(v), D * lq4,
                                                 D * lk4, lq4, lk4, CF._ptr(work), CF._ptr(dscores), CF._ptr(lse), CF._ptr(delta),
                                                 CF._ptr(dq), CF._ptr(dk), CF._ptr(dv), D * lq4, D * lk4, b, H, d, lq4, lk,
                                                 CF._ptr(q_lens), CF._ptr(k_lens), Tp, p_attn, seed_attn, CF._stream()),
                       "csn_varlen_attn_bwd_f32")
        else:
            _lib.check(L.csn_cross_attn_bwd_f32(CF._ptr(datt), CF._ptr(att), D * lq4, CF._ptr(q), CF._ptr(k), CF._ptr(v), D * lq4,
                                                D * lk4, lq4, lk4, CF._ptr(work), CF._ptr(dscores), CF._ptr(lse), CF._ptr(delta),
                                                CF._ptr(dq), CF._ptr(dk), CF._ptr(dv), D * lq4, D * lk4, b, H, d, lq4, lk, Tp,
                                                p_attn, seed_attn, CF._stream()), "csn_cross_attn_bwd_f32")
        del work, dscores
        need = ctx.needs_input_grad
        dq /= temperature                                            # Qs = (xq Wq^T) / sqrt(d)
        dw_q = CF.project_wgrad(dq, xq_cm) if need[3] else None
        dw_k = CF.project_wgrad(dk, xk_cm) if need[4] else None
        dw_v = CF.project_wgrad(dv, xv_cm) if need[5] else None
        dxq = dxk = dxv = None
        if need[0]:
            dxq = CF.project(dq, w_qs.t().contiguous()) + (dz if dz_res is None else dz_res)      # projection + residual
            dxq = dxq[:, :, :lq].transpose(1, 2)
        if need[1]:
            dxk = CF.project(dk, w_ks.t().contiguous())[:, :, :lk].transpose(1, 2)
        if need[2]:
            dxv = CF.project(dv, w_vs.t().contiguous())[:, :, :lk].transpose(1, 2)
        return dxq, dxk, dxv, dw_q, dw_k, dw_v, dw_fc, None, None, None, None, None, None, None, None, None


class ScaledDotProductAttention(nn.Module):
    """MinkowskiNet/models/attention.py:59-73.  (b, H, lq, d) x (b, H, lk, d) x (b, H, lk, d) -> ((b, H, lq, d), attn)."""

    def __init__(self, temperature, attn_dropout=0.1):
        super().__init__()
        self.temperature = temperature
        self.dropout = nn.Dropout(attn_dropout)

    def forward(self, q, k, v):
        # the stand-alone form is not on the layer's hot path (MultiHeadAttention feeds the kernels channel-major data
        # directly); it is served by the same fused path with identity projections folded away
        b, H, lq, d = q.shape
        from .sdpa import sdpa_cross
        return sdpa_cross(q, k, v, float(self.temperature), self.dropout.p if self.training else 0.0)


class ScaledDotProduct(nn.Module):
    """MinkowskiNet/models/attention.py:76-104 for dense inputs: q k^T / temperature (two pooled descriptors: a 1x1 result)."""

    def __init__(self, temperature):
        super().__init__()
        self.temperature = temperature

    def forward(self, q, k):
        if q.ndim == 2:
            q = q.unsqueeze(0)
        if k.ndim == 2:
            k = k.unsqueeze(0)
        return torch.bmm(q, k.permute(0, 2, 1)) / self.temperature

    def __repr__(self):
        return f"{self.__class__.__name__}(temperature={self.temperature})"


class MultiHeadAttention(nn.Module):
    """MinkowskiNet/models/attention.py:9-56: same parameters (w_qs, w_ks, w_vs, fc without bias; LayerNorm eps 1e-6; both
    dropouts p = 0.1 live in train mode), ``forward(q, k, v)`` on (b, len, d_model) point-major features with
    len_q != len_k allowed, returns ``(out (b, len_q, d_model), attn (b, n_head, len_q, len_k))``."""

    def __init__(self, n_head, d_model, d_k, d_v, dropout=0.1, return_attention: bool = True):
        super().__init__()
        self.n_head, self.d_k, self.d_v = n_head, d_k, d_v
        # one kernel head width for Q / K and V / O: d_k != d_v, or a width without a kernel instance (the reference builds
        # d_k = d_v = d_model / n_head, hrnet.py:343 — whatever that is), runs at the next instance with zero-padded weights
        self.d_head = CF.kernel_head_width(max(d_k, d_v))
        self.w_qs = nn.Linear(d_model, n_head * d_k, bias=False)
        self.w_ks = nn.Linear(d_model, n_head * d_k, bias=False)
        self.w_vs = nn.Linear(d_model, n_head * d_v, bias=False)
        self.fc = nn.Linear(n_head * d_v, d_model, bias=False)
        self.attention = ScaledDotProductAttention(temperature=d_k ** 0.5)
        self.dropout = nn.Dropout(dropout)
        self.norm = nn.LayerNorm(d_model, eps=1e-6)
        self.return_attention = return_attention

    def kernel_weights(self):
        """(W_q, W_k, W_v, W_fc) at the kernels' head width (see csn_amd.csa_models.MultiHeadAttention.kernel_weights)."""
        H, d = self.n_head, self.d_head
        F = torch.nn.functional

        def rows(w, dh):
            return w if dh == d else F.pad(w.view(H, dh, -1), (0, 0, 0, d - dh)).reshape(H * d, -1)

        wfc = self.fc.weight
        if self.d_v != d:
            wfc = F.pad(wfc.view(-1, H, self.d_v), (0, d - self.d_v)).reshape(-1, H * d)
        return rows(self.w_qs.weight, self.d_k), rows(self.w_ks.weight, self.d_k), rows(self.w_vs.weight, self.d_v), wfc

    def _temperature(self) -> float:
        return 0.0 if self.d_head == self.d_k else float(self.d_k) ** 0.5

    def forward(self, q, k, v):
        if not q.is_cuda:
            raise _lib.CsnError("csn_amd ops need tensors on the MI355X (cuda) device; there is no CPU path")
        p_attn, p_fc = (self.attention.dropout.p, self.dropout.p) if self.training else (0.0, 0.0)
        lq = q.shape[1]
        ws = self.kernel_weights()
        keep = torch.is_grad_enabled() and any(t.requires_grad for t in (q, k, v) + ws)     # (grad mode is off inside forward)
        xhat, attn = _CrossMHA.apply(q.float(), k.float(), v.float(), *ws, self.n_head, self.d_head, float(p_attn), float(p_fc),
                                     self.return_attention, keep, None, None, self._temperature())
        out = xhat[:, :, :lq].transpose(1, 2) * self.norm.weight + self.norm.bias
        return out, attn

    def forward_varlen(self, qs, ks, vs=None):
        """A RAGGED batch in one launch chain: ``qs[i]`` (n_i, d_model), ``ks[i]`` / ``vs[i]`` (m_i, d_model) point-major features
        of shape pair i (``vs=None``: values = keys), every n_i and m_i different — what MinkowskiNet/models/hrnet.py:378-410,
        456-470 does with one ``forward`` call per shape / shape pair.  Returns the list of outputs (n_i, d_model), equal to
        ``[self(q[None], k[None], v[None])[0][0] for ...]`` (in train mode: other dropout masks).  The pairs are zero-padded to
        the longest one (query counts rounded up to 4) and carried with their length arrays (csn_varlen_attn_*_f32): a short
        pair costs its own size in the attention kernels; only the projections and the LayerNorm see the padding."""
        vs = ks if vs is None else vs
        if not (len(qs) == len(ks) == len(vs)) or not qs:
            raise ValueError("forward_varlen needs equally long, non-empty lists")
        if not qs[0].is_cuda:
            raise _lib.CsnError("csn_amd ops need tensors on the MI355X (cuda) device; there is no CPU path")
        dev, C = qs[0].device, qs[0].shape[-1]
        nq, nk = [int(t.shape[0]) for t in qs], [int(t.shape[0]) for t in ks]
        if any(int(v.shape[0]) != m for v, m in zip(vs, nk)) or min(nq) < 1 or min(nk) < 1:
            raise ValueError("keys and values of a pair must have the same, non-zero length")
        pad = lambda ts, n: torch.stack([torch.nn.functional.pad(t.float(), (0, 0, 0, n - t.shape[0])) for t in ts])
        Lq, Lk = max(nq), max(nk)
        q, k = pad(qs, Lq), pad(ks, Lk)
        v = k if vs is ks else pad(vs, Lk)
        q_lens = torch.tensor([_up(n, 4) for n in nq], dtype=torch.int32, device=dev)
        k_lens = torch.tensor(nk, dtype=torch.int32, device=dev)
        p_attn, p_fc = (self.attention.dropout.p, self.dropout.p) if self.training else (0.0, 0.0)
        ws = self.kernel_weights()
        keep = torch.is_grad_enabled() and any(t.requires_grad for t in tuple(qs) + tuple(ks) + tuple(vs) + ws)
        xhat, _ = _CrossMHA.apply(q, k, v, *ws, self.n_head, self.d_head, float(p_attn), float(p_fc), False, keep, q_lens, k_lens,
                                  self._temperature())
        out = xhat.transpose(1, 2) * self.norm.weight + self.norm.bias                     # (b, Lq4, C)
        return [out[i, :n] for i, n in enumerate(nq)]
