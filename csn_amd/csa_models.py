"""MI355X-native drop-in for the model surface of ``MID-FC/csa_models.py`` (marios2019/CSN).

Same public names, constructor arguments, ``forward`` signatures, ``state_dict`` keys and results as the
reference (``from csa_models import *`` in csa_training.py:18 / ssa_training.py:17), but the attention
path — Q/K/V projections, block-diagonal scaled-dot-product attention, output projection + residual +
LayerNorm, and all of their gradients — runs in hand-written gfx950 kernels behind the C ABI of
``include/csn_hip.h``.  Nothing in this file falls back to eager PyTorch for that path: CPU tensors or a
missing ``libcsn_hip.so`` raise.

Differences a caller can see (all opt-in or strictly more permissive):
  * ``MultiHeadAttention(..., block=500, n_blocks=20)`` exposes the reference's hard-coded chunking
    (csa_models.py:83-84) so other point counts can be run; the defaults reproduce the reference,
    including "points beyond 20*500 are ignored" (N < 10000 raises ``IndexError`` like the reference).
  * outputs of the attention are views of channel-major buffers (values identical).
  * the attention probabilities (2nd return value of ``MultiHeadAttention.forward``; every caller in the
    reference discards it) are only materialised on request (``return_attn=True``), otherwise ``None``.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as CF, tuning
from ._lib import CsnError

__all__ = ["ScaledDotProductAttention", "MultiHeadAttention", "CrossShapeAt", "get_model",
           "backbone_fc_ssa_logit", "backbone_fc_csa_logit", "backbone_ssa_fc_logit", "backbone_csa_fc_logit", "device"]

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")       # csa_models.py:8


def _channel_major(x: torch.Tensor, n_points: int) -> torch.Tensor:
    """(B, C, N, 1) or (B, C, N) -> contiguous (B, C, n_points) on the GPU, fp32."""
    if x.dim() == 4:
        x = x.squeeze(-1)
    if x.shape[-1] < n_points:
        raise IndexError(f"index {n_points - 1} is out of bounds for dimension 2 with size {x.shape[-1]}")
    if not x.is_cuda:
        x = x.to(device, non_blocking=True)
    return x[..., :n_points].to(torch.float32).contiguous()


class _HostNeighbourStack:
    """A (B, K+1, C, N) neighbour stack that is still in host memory, in the form CrossShapeAt._csa_cm_overlapped takes
    (``wait()`` -> the device stack with slot 0 = the shape itself).  The transfer is ONE copy of the whole contiguous tensor
    (the strided view without slot 0 would first be gathered by a single host thread: 170 ms instead of 25 ms for the 1.31 GB
    of config 3), issued from ``wait()`` on a side stream — i.e. after the caller has queued the evaluations that need no
    neighbour data, so that a pageable source, whose staging blocks the host, still overlaps with them on the GPU."""
    reuse_descriptors = False

    def __init__(self, nb: torch.Tensor, xc: torch.Tensor, trust_slot0: bool, streams: dict):
        self._nb, self._xc, self._trust = nb, xc, trust_slot0
        self._streams = streams                      # the owning module's side streams, one per device, made on first use

    def wait(self) -> torch.Tensor:
        dev = self._xc.device
        if dev not in self._streams:
            self._streams[dev] = torch.cuda.Stream(dev)
        side = self._streams[dev]
        main = torch.cuda.current_stream(dev)
        with torch.cuda.stream(side):                # (no wait on `main`: the copy runs beside the evaluations queued there)
            stack = self._nb.to(dev, non_blocking=True).to(torch.float32)
        main.wait_stream(side)
        stack.record_stream(main)
        if not self._trust:
            stack[:, 0] = self._xc                   # the query shape itself (csa_models.py:210, 232)
        return stack


class ScaledDotProductAttention(nn.Module):
    """softmax((q / temperature) k^T) v on (B, H, T, d) tensors (csa_models.py:128-144)."""

    def __init__(self, temperature, attn_dropout=0.1):
        super().__init__()
        self.temperature = temperature
        self.dropout = nn.Dropout(attn_dropout)

    def forward(self, q, k, v):
        from .sdpa import sdpa_block
        return sdpa_block(q, k, v, float(self.temperature), self.dropout.p if self.training else 0.0)


class MultiHeadAttention(nn.Module):
    """Block-diagonal multi-head attention (csa_models.py:37-125)."""

    def __init__(self, n_head, d_model, d_k, d_v, dropout=0.1, block=CF.REF_BLOCK, n_blocks=CF.REF_NBLOCKS, math=None):
        super().__init__()
        # arithmetic of this module's contractions: None = the library's process default (bf16x3 unless csn_set_math_mode
        # changed it), or 'fp32' | 'bf16x3' | 'bf16' | 'fp16' (include/csn_hip.h, CSN_MATH_*) for this module only
        self.math_mode = CF.mode_id(math)
        self.n_head, self.d_k, self.d_v = n_head, d_k, d_v
        # the fused attention kernels run one head width for Q / K and V / O: d_k != d_v (csa_models.py:42 allows it; no
        # caller uses it) or a width without a kernel instance runs at the next instance with zero-padded weights
        self.d_head = CF.kernel_head_width(max(d_k, d_v))
        self.block, self.n_blocks = block, n_blocks
        self.w_qs = nn.Linear(d_model, n_head * d_k, bias=False)
        self.w_ks = nn.Linear(d_model, n_head * d_k, bias=False)
        self.w_vs = nn.Linear(d_model, n_head * d_v, bias=False)
        self.fc = nn.Linear(n_head * d_v, d_model, bias=False)
        self.attention = ScaledDotProductAttention(temperature=d_k ** 0.5)
        self.dropout = nn.Dropout(dropout)
        self.norm = nn.LayerNorm(d_model, eps=CF.LN_EPS)

    # -- helpers ------------------------------------------------------------------------------------------
    def geometry(self, block: Optional[int] = None, n_blocks: Optional[int] = None, n_points: Optional[int] = None) -> CF.MHAGeometry:
        """The chunking of a call.  ``self.n_blocks = None`` (constructor ``n_blocks=None``) means "every point": as many
        blocks as ``n_points`` needs, the last one short when ``n_points`` is not a multiple of ``block`` (a generalisation:
        the reference's 20 x 500 silently drops points beyond 10000 and raises below, csa_models.py:83-90)."""
        block = self.block if block is None else block
        n_blocks = self.n_blocks if n_blocks is None else n_blocks
        if n_blocks is None:
            if n_points is None:
                raise ValueError("n_blocks=None needs the point count of the call")
            n_blocks = (n_points + block - 1) // block
            return CF.MHAGeometry(self.n_head, self.d_head, block, n_blocks, 0 if n_blocks * block == n_points else n_points,
                                  self._temperature())
        return CF.MHAGeometry(self.n_head, self.d_head, block, n_blocks, 0, self._temperature())

    def _temperature(self) -> float:
        return 0.0 if self.d_head == self.d_k else float(self.d_k) ** 0.5       # sqrt(d_k) whatever width the kernels run at

    def kernel_weights(self):
        """(W_q, W_k, W_v, W_fc) at the kernels' head width: the modules' own weights when d_k = d_v = that width, else
        zero rows appended per head to W_q / W_k (d_k -> d_head) and W_v (d_v -> d_head), zero columns per head to W_fc.
        Built with differentiable pads, so the parameters receive the gradients of their real rows."""
        H, d = self.n_head, self.d_head

        def rows(w, dh):
            return w if dh == d else F.pad(w.view(H, dh, -1), (0, 0, 0, d - dh)).reshape(H * d, -1)

        wfc = self.fc.weight
        if self.d_v != d:
            wfc = F.pad(wfc.view(-1, H, self.d_v), (0, d - self.d_v)).reshape(-1, H * d)
        return rows(self.w_qs.weight, self.d_k), rows(self.w_ks.weight, self.d_k), rows(self.w_vs.weight, self.d_v), wfc

    def dropout_rates(self):
        """(attention-probability p, post-fc p): live only in train mode (csa_models.py:133-141, 56, 115)."""
        return (self.attention.dropout.p, self.dropout.p) if self.training else (0.0, 0.0)

    def evaluate(self, x_all: torch.Tensor, plan: CF.EvalPlan, geo: Optional[CF.MHAGeometry] = None, n_head_evals: int = 0,
                 want_sums: bool = False, link_mix: bool = False):
        """Normalised (pre-affine) outputs (E, C, NP) of a batch of evaluations over shared slots
        (n_head_evals > 0: also the leading maps as a second result — a tensor, or with link_mix the LinkedMaps that
        CF.csa_mix takes; want_sums: also their (E, C) sums over the points, see CF.mha_evals)."""
        p_attn, p_fc = self.dropout_rates()
        with CF.math_mode(self.math_mode):
            return CF.mha_evals(x_all, *self.kernel_weights(), plan, geo or self.geometry(), p_attn, p_fc, n_head_evals, want_sums, link_mix)

    def plan(self, kind: str, B: int, K1: int, dev) -> CF.EvalPlan:
        """Cached evaluation plans (slot maps live on the device; building one costs a few small H2D copies)."""
        key = (kind, B, K1, str(dev))
        cache = self.__dict__.setdefault("_plans", {})
        if key not in cache:
            ar = np.arange(B)
            if kind == "self":            # MHA(x, x, x)
                cache[key] = CF.EvalPlan(ar, ar, B, dev)
            elif kind == "cross":         # MHA(x, y, y): slots [x_0..x_B-1, y_0..y_B-1]
                cache[key] = CF.EvalPlan(ar, ar + B, 2 * B, dev)
            elif kind == "qkv":           # three distinct inputs: values live B slots after the keys
                cache[key] = CF.EvalPlan(ar, ar + B, 3 * B, dev, v_shift=B)
            elif kind == "csa":
                # slots s = b*K1 + k (k = 0: the query shape).  Evaluations:
                #   [b*K1 + k]            MHA(x_b, x_bk, x_bk)   (k = 0: self attention)   -> mixed   (csa_models.py:232-238)
                #   [B*K1 + b*(K1-1)+k-1] MHA(x_bk, x_bk, x_bk)  k >= 1: only its mean is used       (csa_models.py:214-220)
                b, k = np.meshgrid(ar, np.arange(K1), indexing="ij")
                mix_q, mix_kv = (b * K1).reshape(-1), (b * K1 + k).reshape(-1)
                nbr = (b * K1 + k)[:, 1:].reshape(-1)
                cache[key] = CF.EvalPlan(np.concatenate((mix_q, nbr)), np.concatenate((mix_kv, nbr)), B * K1, dev)
            elif kind == "csa_train":
                # with dropout live the reference's two self calls (:210 for the pooled descriptor, :232 for the mix)
                # draw different masks, so the pooled self evaluation is a separate, last group [.. + b]
                b, k = np.meshgrid(ar, np.arange(K1), indexing="ij")
                mix_q, mix_kv = (b * K1).reshape(-1), (b * K1 + k).reshape(-1)
                nbr = (b * K1 + k)[:, 1:].reshape(-1)
                own = ar * K1
                cache[key] = CF.EvalPlan(np.concatenate((mix_q, nbr, own)), np.concatenate((mix_kv, nbr, own)), B * K1, dev)
            elif kind == "self2":         # train mode, own shapes only: [b] mixed self, [B + b] pooled self (separate masks)
                cache[key] = CF.EvalPlan(np.concatenate((ar, ar)), np.concatenate((ar, ar)), B, dev)
            elif kind == "csa_cross":
                # the part of "csa" that needs neighbour data: [b*K + k-1] MHA(x_b, x_bk, x_bk) (mixed), then
                # [B*K + b*K + k-1] MHA(x_bk, x_bk, x_bk) (only its mean is used); slots as in "csa"
                b, k = np.meshgrid(ar, np.arange(K1), indexing="ij")
                nbr = (b * K1 + k)[:, 1:].reshape(-1)
                own = (b * K1)[:, 1:].reshape(-1)
                cache[key] = CF.EvalPlan(np.concatenate((own, nbr)), np.concatenate((nbr, nbr)), B * K1, dev)
            elif kind == "cross_only":
                # descriptor reuse (multi-GPU): only [b*K + k-1] MHA(x_b, x_bk, x_bk) — the neighbours' own self-attention is
                # their owners' work (csn_amd.sharding.PendingStack.gather_pooled); slots as in "csa"
                b, k = np.meshgrid(ar, np.arange(K1), indexing="ij")
                nbr = (b * K1 + k)[:, 1:].reshape(-1)
                own = (b * K1)[:, 1:].reshape(-1)
                # Q is read of the own slots b*K1 only, K / V of the neighbour slots b*K1 + k only
                cache[key] = CF.EvalPlan(own, nbr, B * K1, dev, q_ranges=[(0, K1, B)],
                                         kv_ranges=[(k_, K1, B) for k_ in range(1, K1)])
            else:
                raise ValueError(kind)
        return cache[key]

    def affine(self, xhat: torch.Tensor) -> torch.Tensor:
        """LayerNorm's gamma/beta on channel-major activations (csa_models.py:118)."""
        return xhat * self.norm.weight[:, None] + self.norm.bias[:, None]

    @staticmethod
    def _same(a: torch.Tensor, b: torch.Tensor) -> bool:
        return a is b or (a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() == b.stride())

    def _run(self, Q, K, V, geo):
        """Slots: the distinct inputs among (Q, K, V), each projected once.  The reference only ever calls
        (x, x, x) and (x, x_k, x_k) (csa_models.py:205,232,237); three distinct inputs work too."""
        B = Q.shape[0]
        npts = geo.n_points
        xq = _channel_major(Q, npts)
        dev = xq.device
        if self._same(K, V):
            if self._same(Q, K):
                x_all, plan = xq, self.plan("self", B, 1, dev)
            else:
                x_all, plan = torch.cat((xq, _channel_major(K, npts)), dim=0), self.plan("cross", B, 1, dev)
        else:
            x_all = torch.cat((xq, _channel_major(K, npts), _channel_major(V, npts)), dim=0)
            plan = self.plan("qkv", B, 1, dev)
        return self.affine(self.evaluate(x_all, plan, geo))                        # (B, C, NP)

    # -- reference surface ------------------------------------------------------------------------------
    def self_attention(self, x):
        """Unchunked self-attention over all N points (csa_models.py:59-79)."""
        N = x.shape[2]
        y = self._run(x, x, x, self.geometry(block=N, n_blocks=1))
        return y.permute(0, 2, 1), None

    def forward(self, Q, K, V, mode=None, return_attn: bool = False):
        """(B, C, N, 1) x3 -> ((B, n_blocks*block, C), attn-of-last-block or None)   (csa_models.py:81-125)."""
        y = self._run(Q, K, V, self.geometry(n_points=Q.shape[2]))
        attn = None
        if return_attn:
            from .sdpa import last_block_probabilities
            attn = last_block_probabilities(self, Q, K)
        return y.permute(0, 2, 1), attn


class CrossShapeAt(nn.Module):
    """csa_models.py:146-404."""

    def __init__(self, num_classes, d_model, n_heads, K=None, d_k=256, d_v=256, attention_type='ssa',
                 after_fc=False, device=None, block=CF.REF_BLOCK, n_blocks=CF.REF_NBLOCKS, math=None, feature_width=None):
        """Arguments as the reference's (csa_models.py:147).  The reference hard-codes 256 for the widths of ``fc_1``,
        ``logit`` and the compatibility head (:150-151, :160-161) and 20 x 500 for the chunking (:83-84): here they follow
        ``d_model`` / ``block`` / ``n_blocks`` (identical for the defaults), so that the other BASELINE configurations —
        e.g. 8 x 50000 points x 96 channels in 100 blocks — run through the same module.  ``feature_width`` (the width of
        the feature map ``fc_1`` produces and ``logit`` / the compatibility head consume) defaults to ``d_model`` up to the
        widest map this library runs (256) and to the reference's constant 256 beyond it — the reference's only other
        ``d_model`` is the backbone's 928 of ``backbone_{ssa,csa}_fc_logit`` (:406-409, :416-419), whose state dict this
        reproduces key by key."""
        super().__init__()
        self.d_model = d_model
        self.feature_width = fw = feature_width if feature_width is not None else (d_model if d_model <= 256 else 256)
        self.fc_1 = self._conv1x1_bn_relu(928, fw)             # never executed by any forward (csa_models.py:191-202); kept for checkpoints
        self.logit = self._conv1x1(fw, num_classes)
        self.attention = MultiHeadAttention(n_heads, d_model, d_k, d_v, block=block, n_blocks=n_blocks, math=math)
        self.attention_type = attention_type
        self.after_fc = after_fc
        self.device = device
        self.compat_layout = "reference"      # see get_csa_feats
        # CSADatasetK puts the shape itself into slot 0 of neighbor_feats (features_data_loader.py:124); a caller that
        # guarantees it can set this to let the kernels read a device-resident, contiguous neighbour stack in place
        # (no 1.3 GB gather per step at config 3).  Default: slot 0 is taken from x, exactly like the reference (:210, :232).
        self.trust_neighbor_slot0 = False
        self._side_streams = {}               # device -> the stream host-resident neighbour stacks are copied on (made on first use)
        if 'csa' in self.attention_type:
            self.K = K
            self.compatibility_q = nn.Linear(fw, fw)
            self.compatibility_k = nn.Linear(fw, fw)

    @staticmethod
    def _conv1x1(nin, nout, use_bias=False):
        layer = nn.Conv2d(nin, nout, kernel_size=1, stride=1, padding='same', bias=use_bias)
        nn.init.xavier_uniform_(layer.weight)                                  # csa_models.py:179
        return layer

    @classmethod
    def _conv1x1_bn_relu(cls, nin, nout):
        return nn.Sequential(nn.Sequential(cls._conv1x1(nin, nout), nn.BatchNorm2d(nout)), nn.ReLU())

    # -- forward ----------------------------------------------------------------------------------------------
    def forward(self, x, mode=None, neighbor_feats=None):
        with CF.math_mode(self.attention.math_mode):       # the logit layer and the mix run in the module's mode too
            if self.attention_type == 'ssa':
                return self.forward_ssa(x, mode)
            if self.attention_type == 'csa':
                return self.forward_csa(x, neighbor_feats, mode)
        return x                                                               # csa_models.py:182-189 falls through

    def _logits(self, feats_cm: torch.Tensor) -> torch.Tensor:
        """1x1 conv without bias on channel-major features (csa_models.py:151,194,201) -> (B, n_cls, N, 1)."""
        w = self.logit.weight.view(self.logit.weight.shape[0], -1)
        n_cls = w.shape[0]
        pad = (-n_cls) % 4                                 # the HIP GEMMs want row counts that are multiples of 4
        if pad:
            w = F.pad(w, (0, 0, 0, pad))
        return CF.linear_cm(feats_cm.contiguous(), w)[:, :n_cls].unsqueeze(-1)

    def _logits_of_input(self, x) -> torch.Tensor:
        """``after_fc=False`` (csa_models.py:191-202): the attention is skipped and the logit layer is applied to the input as it
        stands — to every point of it (the 20 x 500 chunking belongs to the attention).  The HIP GEMM wants point counts in
        multiples of 4: other counts run with up to three zero points appended, which are cut off again."""
        n = x.shape[2]
        xc = _channel_major(x, n)
        pad = (-n) % 4
        if pad:
            xc = F.pad(xc, (0, pad))
        return self._logits(xc)[:, :, :n]

    def forward_ssa(self, x, mode=None):
        if not self.after_fc:
            return self._logits_of_input(x)
        return self._logits(self._ssa_cm(x))

    def forward_csa(self, x, x_neighbors, mode=None):
        if not self.after_fc:
            return self._logits_of_input(x)
        return self._logits(self._csa_cm(x, x_neighbors))

    def _ssa_cm(self, x) -> torch.Tensor:
        att = self.attention
        geo = att.geometry(n_points=x.shape[2])
        xc = _channel_major(x, geo.n_points)
        return att.affine(att.evaluate(xc, att.plan("self", xc.shape[0], 1, xc.device), geo))

    def get_ssa_feats(self, x, mode=None):
        """(B, 256, N, 1) -> ((B, 256, N', 1), None)   (csa_models.py:204-207)."""
        return self._ssa_cm(x).unsqueeze(-1), None

    def _csa_cm(self, x, x_neighbors, return_parts: bool = False):
        """Cross-shape attention features, channel-major (B, C, NP)   (csa_models.py:209-242).

        Slots: s = b*(K+1) + k holds x_b (k = 0) or its k-th neighbour.  Evaluations, in this order:
          [b*(K+1) + k]          k = 0: SSA(x_b) = MHA(x_b, x_b, x_b);  k >= 1: MHA(x_b, x_bk, x_bk)   (mixed, :232-238)
          [B*(K+1) + b*K + k-1]  SSA(x_bk), only its mean over points is used                          (:214-220)
        In eval mode the reference's two self calls (:210 and :232) are the same numbers; they are computed once.
        """
        att = self.attention
        geo = att.geometry(n_points=x.shape[2])
        npts = geo.n_points
        xc = _channel_major(x, npts)                                           # (B, C, NP)
        B, C, _ = xc.shape
        if callable(getattr(x_neighbors, "wait", None)):
            return self._csa_cm_overlapped(xc, x_neighbors, return_parts)
        K1 = x_neighbors.shape[1]
        K = K1 - 1
        dev = xc.device
        nb = x_neighbors
        if nb.dim() == 5:
            nb = nb.squeeze(-1)
        if nb.shape[-1] < npts:
            raise IndexError(f"index {npts - 1} is out of bounds for dimension 2 with size {nb.shape[-1]}")
        if (self.trust_neighbor_slot0 and nb.is_cuda and nb.dtype == torch.float32 and nb.is_contiguous()
                and nb.shape[-1] == npts):
            x_all = nb.view(B * K1, C, npts)                                   # slot 0 already holds the shape itself
        elif not nb.is_cuda and nb.is_contiguous() and nb.shape[-1] == npts and K > 0:
            # the stack arrives on the CPU (csa_training.py:198-202, csa_models.py:216): ONE transfer of the whole contiguous
            # tensor on a side stream, under the self-attention of the query shapes, which needs no neighbour data
            return self._csa_cm_overlapped(xc, _HostNeighbourStack(nb, xc, self.trust_neighbor_slot0, self._side_streams), return_parts)
        else:
            x_all = torch.empty((B, K1, C, npts), device=dev, dtype=torch.float32)
            x_all[:, 0] = xc                                                   # the query shape itself (:210, :232)
            if K > 0:
                x_all[:, 1:] = nb[:, 1:, :, :npts].to(dev, non_blocking=True)  # (device tensors of another layout / dtype)
            x_all = x_all.view(B * K1, C, npts)

        train = any(r > 0 for r in att.dropout_rates())
        E1, E2 = B * K1, B * K
        # xhat: all (E, C, NP) maps, used only through their means; xhat_mix: the E1 maps that are mixed (same storage)
        xhat, xhat_mix, sums = att.evaluate(x_all, att.plan("csa_train" if train else "csa", B, K1, dev), geo,
                                            n_head_evals=E1, want_sums=True, link_mix=True)
        gamma, beta = att.norm.weight, att.norm.bias
        # pooled descriptors y_k = mean_n SSA(x_k)  (:211-212, :218-219); the affine commutes with the mean.  The sums over
        # the points come out of the out-projection's epilogue (no pass over the maps)
        means = sums / npts                                                    # (E, C)
        own = means[E1 + E2:].view(B, 1, C) if train else means[:E1].view(B, K1, C)[:, :1]
        pooled_hat = torch.cat((own, means[E1:E1 + E2].view(B, K, C)), dim=1)
        pooled = pooled_hat * gamma + beta                                     # (B, K+1, C)
        comp = self._compatibility(pooled)                                     # (B, K+1)
        feats = CF.csa_mix(xhat_mix, comp, gamma, beta, B, K1)                   # sum_k comp_k * affine(xhat_k)  (:233, :238)
        return (feats, comp, pooled) if return_parts else feats

    def _csa_cm_overlapped(self, xc, pending, return_parts: bool = False):
        """The same features when the neighbour stack is still on its way (``pending.wait()`` returns the (B, K+1, C, N[, 1])
        device stack with slot 0 = the shape itself; csn_amd.sharding.PendingStack): the evaluations that need only the
        query shapes — a quarter of the work at K = 3 — run first, under the exchange; everything that reads neighbour data
        follows the wait.  Same arithmetic as ``_csa_cm`` (in train mode the dropout masks are drawn in a different order)."""
        att = self.attention
        geo = att.geometry(n_points=xc.shape[2])
        npts = geo.n_points
        B, C, _ = xc.shape
        dev = xc.device
        train = any(r > 0 for r in att.dropout_rates())
        xc = xc.contiguous()
        # phase 1 (no neighbour data): [b] mixed self, and in train mode [B + b] the pooled self with its own masks
        xh1, head1, s1 = att.evaluate(xc, att.plan("self2" if train else "self", B, 1, dev), geo, n_head_evals=B, want_sums=True,
                                       link_mix=True)
        nb = pending.wait()
        if nb.dim() == 5:
            nb = nb.squeeze(-1)
        K1 = nb.shape[1]
        K = K1 - 1
        if not (nb.is_cuda and nb.dtype == torch.float32 and nb.is_contiguous() and nb.shape[-1] == npts):
            raise ValueError("a pending neighbour stack must resolve to a contiguous fp32 device tensor (B, K+1, C, n_points)")
        x_all = nb.view(B * K1, C, npts)
        gamma, beta = att.norm.weight, att.norm.bias
        m1 = s1 / npts
        own = m1[B:] if train else m1[:B]
        if getattr(pending, "reuse_descriptors", False) and K > 0:
            # descriptor reuse: the pooled SSA descriptor of a neighbour is its OWNER's own descriptor, fetched through one small
            # differentiable all-gather — the K neighbour self-attention evaluations per shape (:214-220) are not run here
            xh2, head2 = att.evaluate(x_all, att.plan("cross_only", B, K1, dev), geo, n_head_evals=B * K, link_mix=True)
            nbr_hat = pending.gather_pooled(own)                                        # (B, K, C), pre-affine means
        else:
            # phase 2: [b*K + k-1] cross evaluations (mixed), [B*K + b*K + k-1] neighbour self-attention (pooled only)
            xh2, head2, s2 = att.evaluate(x_all, att.plan("csa_cross", B, K1, dev), geo, n_head_evals=B * K, want_sums=True,
                                           link_mix=True)
            nbr_hat = (s2 / npts)[B * K:].view(B, K, C)
        pooled_hat = torch.cat((own.view(B, 1, C), nbr_hat), dim=1)
        pooled = pooled_hat * gamma + beta
        comp = self._compatibility(pooled)
        feats = CF.csa_mix(head2, comp, gamma, beta, B, K1, xself=head1)   # own maps and cross maps stay where they are
        return (feats, comp, pooled) if return_parts else feats

    def _compatibility(self, pooled: torch.Tensor) -> torch.Tensor:
        """softmax_k <normalize(Wq y_0 + b), normalize(Wk y_k + b)>   (csa_models.py:222-230).

        ``compat_layout == "reference"`` reproduces the reference's row bookkeeping for B > 1: the key
        descriptors are concatenated neighbour-major (rows k*B + b, :213,:220) and then re-viewed as
        (B, K+1, C) (:227).  ``"per_shape"`` scores every shape against its own neighbours."""
        B, K1, C = pooled.shape
        f32 = all(t.dtype == torch.float32 for t in (pooled, self.compatibility_q.weight, self.compatibility_q.bias,
                                                     self.compatibility_k.weight, self.compatibility_k.bias))
        if pooled.is_cuda and f32 and tuning.current().fused_compat_head and C <= 256 and K1 <= 8:
            # one launch forward, two backward (csn_compat_fwd_f32 / _bwd_f32) instead of ~45 small library launches per step
            return CF.compat_head(pooled, self.compatibility_q.weight, self.compatibility_q.bias, self.compatibility_k.weight,
                                  self.compatibility_k.bias, self.compat_layout == "reference")
        u_q = F.normalize(self.compatibility_q(pooled[:, 0]), dim=-1)
        keys = pooled.transpose(0, 1).reshape(K1 * B, C).view(B, K1, C) if self.compat_layout == "reference" else pooled
        u_k = F.normalize(self.compatibility_k(keys), dim=-1)
        return F.softmax(torch.einsum("bc,bkc->bk", u_q, u_k), dim=-1)

    def get_csa_feats(self, x, x_neighbors, mode=None):
        return self._csa_cm(x, x_neighbors).unsqueeze(-1)

    # -- shape-graph construction (csa_models.py:244-404) -----------------------------------------------
    def get_retrieval_measure(self, ssa_feats_1, ssa_feats_2):
        """(S1, N, C), (S2, N, C) point-major SSA features -> (S1, S2) retrieval scores."""
        f1 = ssa_feats_1.to(device, torch.float32)
        f2 = ssa_feats_2.to(device, torch.float32)
        return CF.retrieval_measure(f1, f2)

    def get_knn_graph(self, ssa_feats_1, ssa_feats_2, K):
        scores, knn_graph = self.get_retrieval_measure(ssa_feats_1, ssa_feats_2).topk(K + 1, -1)
        return knn_graph

    def get_all_feats(self, logs_dir, train_dataloader, K, mode):
        chunks = []
        for feats, label in train_dataloader:
            feats = torch.squeeze(feats, dim=1)
            with torch.no_grad():
                chunks.append(self._ssa_cm(feats).permute(0, 2, 1).cpu())      # point-major, on the host like :295
        return torch.cat(chunks, dim=0)

    def get_center_shape_indices(self, train_loader):
        """k-means seeding of the candidate set for big categories (csa_models.py:302-332)."""
        from sklearn.cluster import KMeans
        glob = []
        for feats, label in train_loader:
            feats = torch.squeeze(feats, dim=1)
            with torch.no_grad():
                glob.append(torch.amax(self._ssa_cm(feats), dim=2))            # max over points -> (B, C)
        glob = torch.cat(glob, dim=0).cpu().numpy()
        n_centers = len(glob) // 10
        kmeans = KMeans(n_clusters=n_centers, random_state=0, n_init=10).fit(glob)
        centers = np.expand_dims(kmeans.cluster_centers_, axis=1)
        return np.argmin(np.sum((centers - glob) ** 2, axis=-1), axis=-1)

    def get_candidate_ssa_feats(self, data_loader, candidate_shape_indices):
        out, counter = [], 0
        for i, (feats, label) in enumerate(data_loader):
            if i != candidate_shape_indices[counter]:
                continue
            feats = torch.squeeze(feats, dim=1)
            with torch.no_grad():
                out.append(self._ssa_cm(feats).permute(0, 2, 1))
            counter += 1
            if counter == len(candidate_shape_indices):
                break
        return torch.cat(out, dim=0)

    def get_retrieval_measure_big(self, query_loader, candidate_loader, candidate_shape_indices):
        candidate_shape_indices.sort()
        cand = self.get_candidate_ssa_feats(candidate_loader, candidate_shape_indices).contiguous()
        rows = []
        for feats, label in query_loader:
            feats = torch.squeeze(feats, dim=1)
            with torch.no_grad():
                f1 = self._ssa_cm(feats).permute(0, 2, 1).contiguous()
                rows.append(CF.retrieval_measure(f1, cand))
        return torch.cat(rows, dim=0)

    def get_knn_graph_big(self, query_loader, candidate_loader, candidate_shape_indices, K):
        measure = self.get_retrieval_measure_big(query_loader, candidate_loader, candidate_shape_indices)
        scores, knn_graph = measure.topk(K + 1, -1)
        return knn_graph


def backbone_ssa_fc_logit(num_classes, n_heads):
    """csa_models.py:406-409: attention width 928 (the backbone's concatenated map), ``after_fc=False`` — the forward is the
    logit layer on a 256-channel map (:191-195); the 928-wide attention only contributes its state-dict keys."""
    return CrossShapeAt(num_classes, 928, n_heads, attention_type='ssa', after_fc=False)


def backbone_csa_fc_logit(num_classes, n_heads, K):
    """csa_models.py:416-419 (see backbone_ssa_fc_logit)."""
    return CrossShapeAt(num_classes, 928, n_heads, K, attention_type='csa', after_fc=False)


def backbone_fc_ssa_logit(num_classes, n_heads, **geometry):
    return CrossShapeAt(num_classes, geometry.pop("d_model", 256), n_heads, attention_type='ssa', after_fc=True, **geometry)


def backbone_fc_csa_logit(num_classes, n_heads, K, **geometry):
    return CrossShapeAt(num_classes, geometry.pop("d_model", 256), n_heads, K, attention_type='csa', after_fc=True, **geometry)


def get_model(attention_type, num_classes, n_heads, K=None, **geometry):
    """csa_models.py:426-432.  ``geometry`` (all optional, the defaults are the reference's constants): d_model=256,
    d_k=256, d_v=256, block=500, n_blocks=20 — e.g. ``get_model('csa', 39, 1, 4, d_model=96, d_k=96, d_v=96,
    n_blocks=100)`` is BASELINE.json's 50000-point, 96-channel configuration — and ``math`` = 'fp32' | 'bf16x3' | 'bf16' |
    'fp16' (default: the library's process default, bf16x3): the arithmetic of this model's contractions."""
    unknown = set(geometry) - {"d_model", "d_k", "d_v", "block", "n_blocks", "math"}
    if unknown:
        raise TypeError(f"get_model: unexpected arguments {sorted(unknown)}")
    if attention_type == 'ssa':
        return backbone_fc_ssa_logit(num_classes, n_heads, **geometry)
    if attention_type == 'csa':
        return backbone_fc_csa_logit(num_classes, n_heads, K, **geometry)
    raise AttributeError(f'{attention_type} not supported')
