"""Autograd wrappers over the C ABI (include/csn_hip.h) for the cross-shape-attention hot path.

torch is plumbing here: it owns device memory, the stream and the autograd tape.  All arithmetic of
the attention path runs in libcsn_hip.so; there is no eager fallback (missing library / CPU tensors
raise).

Vocabulary: a *slot* is one shape's channel-major feature map ``[C][N]``; an *evaluation* is one
``MultiHeadAttention.forward(x_q, x_kv, x_kv)`` of the reference (MID-FC/csa_models.py:81-125), given
as a (query slot, key/value slot) pair.  One CSA forward of a query shape with K neighbours is 2K+1
evaluations over K+1 slots (csa_models.py:209-242); each slot is projected to Q/K/V exactly once.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib, tuning

REF_BLOCK = 500       # MID-FC/csa_models.py:84
REF_NBLOCKS = 20      # MID-FC/csa_models.py:83
LN_EPS = 1e-6         # MID-FC/csa_models.py:57
RESCALE_THRESHOLD = 8.0      # the running softmax maximum is re-based when it grows by more than this (csn_hip.h (2))


def draw_seeds(n: int):
    """n 62-bit dropout seeds from torch's CPU generator (``torch.manual_seed`` reproduces a step).  In a torch.distributed
    job the rank is folded in, so ranks seeded alike (``torch.manual_seed(0)`` everywhere) still draw different masks."""
    seeds = torch.randint(0, 2 ** 62, (n,)).tolist()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        salt = (torch.distributed.get_rank() * 0x9E3779B97F4A7C15) & (2 ** 62 - 1)
        seeds = [s ^ salt for s in seeds]
    return seeds


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_f32(*ts, also=()):
    for t in ts:
        if t is not None and t.dtype not in (torch.float32, torch.int32) and t.dtype not in also:
            raise _lib.CsnError(f"csn_amd ops take fp32 tensors (got {t.dtype}): convert with .float() — nothing here casts silently")


def _need_cuda(*ts, also=()):
    """The one gate in front of the raw-pointer calls: fp32 (int32 for index arrays) tensors on the device, nothing else — the
    library reads its operands as fp32 words, so a half or double tensor would be reinterpreted, not converted (and a half
    buffer read as fp32 runs past its allocation).  ``also``: the 16-bit map types a call site legitimately hands over
    (bf16 gradient maps, fp16 normalised maps of the 16-bit exchange).  Types are checked before devices."""
    _need_f32(*ts, also=also)
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.CsnError("csn_amd ops need tensors on the MI355X (cuda) device; there is no CPU path")


KERNEL_HEAD_WIDTHS = (32, 64, 96, 128, 256)        # instances of the fused attention kernels (csrc/csn_capi.hip dim_ok)


def kernel_head_width(d: int) -> int:
    """The narrowest head width the attention kernels are instantiated for that holds d channels.  Narrower heads run at
    that width with zero channels appended (zero rows of W_q / W_k / W_v, zero columns of W_fc): scores, outputs and the
    gradients of the real rows are unchanged."""
    for w in KERNEL_HEAD_WIDTHS:
        if d <= w:
            return w
    raise ValueError(f"head width {d} exceeds the widest attention kernel ({KERNEL_HEAD_WIDTHS[-1]})")


@dataclass(frozen=True)
class MHAGeometry:
    n_head: int
    d_head: int            # head width the kernels run at (= d_k = d_v in every call of the reference, csa_models.py:147)
    block: int             # points per attention block (csa_models.py:84)
    n_blocks: int          # csa_models.py:83
    n_total: int = 0       # points per shape when the row ends INSIDE the last block (a ragged last block); 0: block * n_blocks
    temperature: float = 0.0   # softmax temperature; 0: sqrt(d_head) (csa_models.py:54).  Set when d_head is a kernel width that
                               # zero-padded projection weights fill (d_k != d_v, or a d_k the kernels have no instance for)

    @property
    def n_points(self) -> int:
        """points per shape that take part: n_blocks full blocks, or n_total with a short last block"""
        return self.n_total or self.block * self.n_blocks

    @property
    def n_padded(self) -> int:
        """n_blocks * block: the layout of the per-point statistics (lse, delta) and of the score blocks"""
        return self.block * self.n_blocks

    @property
    def d_inner(self) -> int:
        return self.n_head * self.d_head

    @property
    def score_pitch(self) -> int:
        return (self.block + 31) // 32 * 32


class EvalPlan:
    """Host-side description of a batch of evaluations over shared slots (the shape graph of one step).

    ``q_slots[e]`` / ``kv_slots[e]``: which slot evaluation e takes its queries / keys+values from
    (values from slot ``kv_slots[e] + v_shift``).  Evaluations that share an output slot in the backward
    pass (the query shape's Q serves K+1 evaluations, a neighbour's K/V two) are split into *colours*: within
    a colour every output slot occurs once, so a colour is one launch that adds into the per-slot gradient
    maps without atomics and without per-evaluation temporaries."""

    def __init__(self, q_slots, kv_slots, n_slots: int, device, v_shift: int = 0, q_ranges=None, kv_ranges=None):
        """q_ranges / kv_ranges (optional): the slots whose Q / K,V projections the evaluations actually read, as arithmetic
        progressions [(first, step, count), ...] — a plan that reads Q of every (K+1)-th slot only (descriptor reuse: the own
        shapes) and K / V of the others then projects, and contracts weight gradients over, just those.  None = every slot."""
        import numpy as np
        q = np.asarray(q_slots, dtype=np.int64).reshape(-1)
        kv = np.asarray(kv_slots, dtype=np.int64).reshape(-1)
        assert q.shape == kv.shape and q.size > 0
        assert q.min() >= 0 and max(q.max(), kv.max() + v_shift) < n_slots
        self.E, self.S, self.v_shift = int(q.size), int(n_slots), int(v_shift)
        self.q_ranges, self.kv_ranges = q_ranges, kv_ranges
        if q_ranges is not None or kv_ranges is not None:
            assert v_shift == 0 and q_ranges is not None and kv_ranges is not None
            cover = lambda rs: set(int(f + st * i) for f, st, c in rs for i in range(c))
            # exactly the slots the plan reads: the ranged backward neither clears nor skips gradient maps, so a covered slot
            # that no evaluation writes would put uninitialised memory into the weight gradients
            assert set(q.tolist()) == cover(q_ranges) and set(kv.tolist()) == cover(kv_ranges), \
                "slot ranges must be exactly the slots the plan reads"
        as_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device)
        self.q_slots, self.kv_slots, self.v_slots = as_dev(q), as_dev(kv), as_dev(kv + v_shift)
        self.dq_colors = [as_dev(c) for c in self._colors(q)]
        self.dkv_colors = [as_dev(c) for c in self._colors(kv)]
        # the same sharing as GROUPS: evaluations ordered by key/value slot (biggest groups first: long work-groups start
        # early), group g = kv_group_items[kv_group_off[g] : kv_group_off[g+1]] — one grouped dK / dV call accumulates a
        # group's products in registers and writes its slot once (csn_block_attn_bwd_dkv_f32, group_offsets)
        items, off = self._groups(kv)
        self._check_groups(items, off, kv)
        self.kv_group_items, self.kv_group_off, self.n_kv_groups = as_dev(items), as_dev(off), int(off.size - 1)
        items, off = self._groups(q)
        self._check_groups(items, off, q)
        self.q_group_items, self.q_group_off, self.n_q_groups = as_dev(items), as_dev(off), int(off.size - 1)
        # the first colour of a pass holds the first evaluation of EVERY slot the pass writes, so it may overwrite; only the
        # gradient maps of slots a pass never writes need a zero fill (the weight gradients read all of them)
        all_slots = np.arange(n_slots)
        self.q_unwritten = as_dev(np.setdiff1d(all_slots, q)).long()
        self.k_unwritten = as_dev(np.setdiff1d(all_slots, kv)).long()
        self.v_unwritten = as_dev(np.setdiff1d(all_slots, kv + v_shift)).long()

    @staticmethod
    def _groups(slots):
        import numpy as np
        order = np.argsort(slots, kind="stable")
        _, start, count = np.unique(slots[order], return_index=True, return_counts=True)
        by_size = np.argsort(-count, kind="stable")
        items = np.concatenate([order[start[g]:start[g] + count[g]] for g in by_size])
        off = np.concatenate(([0], np.cumsum(count[by_size])))
        return items, off

    @staticmethod
    def _check_groups(items, off, slots):
        """The grouped kernels trust these arrays (include/csn_hip.h: a group that mixed slots would silently run every item on
        the first item's operand, items beyond the last offset would never be written): hold the invariants where the arrays
        are made — every evaluation in exactly one group, the offsets cover all of them, one slot per group."""
        import numpy as np
        E = slots.size
        assert items.shape == (E,) and np.array_equal(np.sort(items), np.arange(E)), "group items must be a permutation of the evaluations"
        assert off[0] == 0 and off[-1] == E and (np.diff(off) > 0).all(), "group offsets must cover every evaluation"
        first = np.repeat(slots[items[off[:-1]]], np.diff(off))
        assert np.array_equal(slots[items], first), "a group must hold the evaluations of ONE slot"

    @staticmethod
    def _colors(slots):
        import numpy as np
        seen, rank = {}, np.zeros(slots.size, dtype=np.int64)
        for e, s in enumerate(slots.tolist()):
            rank[e] = seen.get(s, 0)
            seen[s] = rank[e] + 1
        return [np.nonzero(rank == c)[0] for c in range(int(rank.max()) + 1)]


# ------------------------------------------------------------------------------------------------------
# raw (non-differentiable) calls — thin, typed views of the C ABI
# ------------------------------------------------------------------------------------------------------
MATH_MODES = {"fp32": 0, "bf16x3": 1, "bf16": 2, "fp16": 3}      # CSN_MATH_* of include/csn_hip.h


def mode_id(mode) -> Optional[int]:
    """'fp32' | 'bf16x3' | 'bf16' | 'fp16' | 0..3 | None -> the library's mode number (None stays None = no override)."""
    if mode is None:
        return None
    m = MATH_MODES[mode] if isinstance(mode, str) else int(mode)
    if m not in (0, 1, 2, 3):
        raise ValueError(f"unknown math mode {mode!r}")
    return m


def current_mode() -> int:
    """The math mode in effect for the calling thread (process default unless a ``math_mode`` block is open)."""
    return _lib.lib().csn_get_math_mode()


def backward_mode(mode: int) -> int:
    """The mode a backward pass runs in: fp16 is forward-only (gradients of this path reach 1e-7 and underflow fp16), its
    backward runs in bf16."""
    return 2 if mode == 3 else mode


class math_mode:
    """``with math_mode(m):`` — the calling thread's library calls run in mode ``m`` inside the block (csn_set_thread_math_mode;
    other threads are untouched; None = leave as is).  The autograd functions of this file record the mode of their forward and
    open the same block (in ``backward_mode``) in their backward, which runs on autograd's own threads."""

    def __init__(self, mode):
        self.mode = mode_id(mode)

    def __enter__(self):
        if self.mode is not None:
            L = _lib.lib()
            self.prev = L.csn_get_thread_math_mode()          # the thread's own override as it stands (-1: none) — restored
            _lib.check(L.csn_set_thread_math_mode(self.mode))   # exactly, also one set directly through the C ABI
        return self

    def __exit__(self, *exc):
        if self.mode is not None:
            _lib.lib().csn_set_thread_math_mode(self.prev)
        return False


class act16:
    """``with act16(fmt):`` — the calling thread's library calls exchange their intermediate maps as 16-bit planes inside the
    block (csn_set_thread_act16; include/csn_hip.h "16-BIT ACTIVATION MAPS"): fmt 1 = the forward's maps are bf16, 2 = fp16,
    0 = fp32 maps (the default)."""

    def __init__(self, fmt: int):
        self.fmt = int(fmt)

    def __enter__(self):
        L = _lib.lib()
        self.prev = L.csn_get_thread_act16()
        _lib.check(L.csn_set_thread_act16(self.fmt))
        return self

    def __exit__(self, *exc):
        _lib.lib().csn_set_thread_act16(self.prev)
        return False


class score_layout:
    """Bracket: the block-attention calls inside store their scores / P / dS planes tile-major (csn_set_thread_score_layout)."""

    def __init__(self, layout: int):
        self.layout = layout

    def __enter__(self):
        L = _lib.lib()
        self.prev = L.csn_get_thread_score_layout()
        _lib.check(L.csn_set_thread_score_layout(self.layout))
        return self

    def __exit__(self, *exc):
        _lib.lib().csn_set_thread_score_layout(self.prev)
        return False


def _score_flow(mode: int, d: int, T: int) -> int:
    """The attention backward data flow (tuning.KEEP_SCORES / RECOMPUTE_DQ / FLASH) for a forward in `mode` at head width d and
    block T: the configured flow of the mode the BACKWARD runs in, where the library has kernels for it."""
    want = tuning.current().flow_for(backward_mode(mode), d)
    if want == tuning.KEEP_SCORES or mode == 0:
        return tuning.KEEP_SCORES
    with math_mode(backward_mode(mode)):
        bits = _lib.lib().csn_attn_bwd_grouping(d, T)
    if not (bits & 4) or not (bits & 1):
        return tuning.KEEP_SCORES
    if want == tuning.FLASH and not (bits & 8):
        return tuning.RECOMPUTE_DQ
    return want


def fast_math() -> bool:
    """True when the contractions run on the 16-bit matrix-core kernels (math modes 1..3)."""
    return current_mode() != 0


def planes() -> int:
    """Planes of a tile-plane tensor in the current mode: hi + lo in bf16x3, one in the single-product modes."""
    return 2 if current_mode() == 1 else 1


def project(x: torch.Tensor, w: torch.Tensor, div_rows: int = 0, temperature: float = 1.0,
            n_points: Optional[int] = None, split: bool = False) -> torch.Tensor:
    """x (S, C, N), w (R, C)  ->  (S, R, n_points) = w @ x[s]; rows < div_rows divided by temperature.
    split=True (fast math only): the result is returned as bf16 planes (S, 2, R, n_points), x = hi + lo."""
    _need_f32(w)
    _need_cuda(x, also=(torch.bfloat16,))                          # (a bf16 gradient map of math mode 2)
    _need_cuda(w)
    S, C, N = x.shape
    R = w.shape[0]
    npts = N if n_points is None else n_points
    assert x.stride(2) == 1 and x.stride(1) == N and w.is_contiguous()
    x16 = 16 if x.dtype == torch.bfloat16 else 0                   # a bf16 map as input (math mode 2: the bf16 gradient maps)
    assert not (split and x16)
    if split:
        out = torch.empty((S, 2, R, npts), device=x.device, dtype=torch.bfloat16)
        _lib.check(_lib.lib().csn_project_f32(_ptr(x), x.stride(0), N, _ptr(w), R, C, _ptr(out), 2 * R * npts, npts, S, npts,
                                              div_rows, float(temperature), 1, R * npts, _stream()), "csn_project_f32")
        return out
    out = torch.empty((S, R, npts), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().csn_project_f32(_ptr(x), x.stride(0), N, _ptr(w), R, C, _ptr(out), R * npts, npts, S, npts,
                                          div_rows, float(temperature), x16, 0, _stream()), "csn_project_f32")
    return out


def project_wgrad(dout: torch.Tensor, x: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """dout (S, R, NP), x (S, C, N>=NP) -> dw (R, C) = scale * sum_s dout[s] @ x[s][:, :NP]^T."""
    _need_f32(x)
    _need_cuda(dout, also=(torch.bfloat16,))                       # (bf16 gradient maps of the 16-bit exchange)
    _need_cuda(x)
    S, R, NP = dout.shape
    C, N = x.shape[1], x.shape[2]
    assert dout.is_contiguous() and x.stride(2) == 1 and x.stride(1) == N
    dw = torch.empty((R, C), device=x.device, dtype=torch.float32)
    ws_n = _lib.lib().csn_wgrad_workspace_floats(R, C, S, NP)
    ws = torch.empty((ws_n,), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().csn_project_wgrad_f32(_ptr(dout), R * NP, NP, _ptr(x), x.stride(0), N, _ptr(dw), R, C, S, NP,
                                                float(scale), 0, _ptr(ws), ws_n, _stream()), "csn_project_wgrad_f32")
    return dw


def retrieval_measure(f1: torch.Tensor, f2: torch.Tensor, pair_budget: int = 2 ** 28) -> torch.Tensor:
    """f1 (S1, N1, C), f2 (S2, N2, C) point-major SSA features -> (S1, S2) mean-of-max cosine
    (get_retrieval_measure, MID-FC/csa_models.py:244-267).  The per-point maxima scratch is S1*S2*N1 floats — O(S^2) — so
    the query shapes are scored in row chunks of at most ``pair_budget`` scratch floats (1 GiB by default; Vase, 741 x 741
    shapes of 10000 points, would otherwise ask for 22 GB).  Every (query, candidate) score is the same arithmetic whatever
    the chunking."""
    _need_cuda(f1, f2)
    f1 = f1.contiguous()
    f2 = f2.contiguous()
    S1, N1, C = f1.shape
    S2, N2, _ = f2.shape
    out = torch.empty((S1, S2), device=f1.device, dtype=torch.float32)
    rows = max(1, min(S1, pair_budget // max(1, S2 * N1)))
    ws_n = rows * N1 + S2 * N2 + rows * S2 * N1
    ws = torch.empty((ws_n,), device=f1.device, dtype=torch.float32)
    for i in range(0, S1, rows):
        r = min(rows, S1 - i)
        _lib.check(_lib.lib().csn_retrieval_measure_f32(_ptr(f1[i:i + r]), _ptr(f2), _ptr(out[i:i + r]), r, N1, S2, N2, C,
                                                        _ptr(ws), ws_n, _stream()), "csn_retrieval_measure_f32")
    return out


# ------------------------------------------------------------------------------------------------------
# the differentiable unit: a batch of MHA evaluations over shared, once-projected slots
# ------------------------------------------------------------------------------------------------------
class _MHAEvals(torch.autograd.Function):
    """xhat[e] = LayerNorm_noaffine( fc( BlockAttn(Wq x[q_e] / sqrt(d), Wk x[kv_e], Wv x[kv_e]) ) + x[q_e] ).

    Inputs : x_all (S, C, NP) channel-major slots, w_qs/w_ks/w_vs (H*d, C), fc (C, H*d),
             q_slots / kv_slots int32 (E,) on the device.
    Output : xhat (E, C, NP).  The LayerNorm affine is applied by the caller (it is plain elementwise torch,
             and the backward needs the un-affined activations anyway).
    """

    @staticmethod
    def forward(ctx, x_all, w_qs, w_ks, w_vs, w_fc, plan: EvalPlan, geo: MHAGeometry, keep_scores: bool,
                p_attn: float = 0.0, p_fc: float = 0.0, n_head_evals: int = 0, want_sums: bool = False, link=None):
        _need_cuda(x_all, w_qs, w_ks, w_vs, w_fc)
        ctx.link = link
        ctx.set_materialize_grads(False)               # an unused output must not cost a zero-filled (E, C, NP) gradient
        mode = current_mode()
        if mode >= 2 and geo.block > 512:
            mode = 1                                   # the single-product kernels take K / V as tile planes (blocks <= 512 keys)
        ctx.mode = mode
        # the switches as they stand NOW: the backward (another thread, possibly after an override() block has ended) reads this
        # snapshot, never the live object — forward and backward of one step always agree (tuning.override is not thread-safe)
        ctx.tune = tuning.current()
        # 16-bit activation maps: Qs, Ctx, xhat (and in the backward dZ, dCtx) travel between the launches as one 16-bit plane.
        # Only where nothing outside this file reads the maps: the linked form (the mix and the pooled sums consume xhat in
        # kernels, gradients arrive through the link), tile-plane K / V
        tiles_ok = tuning.current().kv_tiles and geo.block <= 512
        ctx.a16 = (mode - 1) if (mode >= 2 and tuning.current().act16 and tiles_ok and link is not None and keep_scores) else 0
        with math_mode(mode), act16(ctx.a16):
            return _MHAEvals._forward(ctx, x_all, w_qs, w_ks, w_vs, w_fc, plan, geo, keep_scores, p_attn, p_fc, n_head_evals,
                                      want_sums, link)

    @staticmethod
    def _forward(ctx, x_all, w_qs, w_ks, w_vs, w_fc, plan, geo, keep_scores, p_attn, p_fc, n_head_evals, want_sums, link):
        # dropout masks are counter-based: two 62-bit seeds from torch's CPU generator (torch.manual_seed reproduces them)
        seed_attn, seed_fc = draw_seeds(2) if (p_attn > 0 or p_fc > 0) else (0, 0)
        q_slots, kv_slots, v_shift = plan.q_slots, plan.kv_slots, plan.v_shift
        L = _lib.lib()
        S, C, NP = x_all.shape
        H, d, D = geo.n_head, geo.d_head, geo.d_inner
        assert NP == geo.n_points and x_all.is_contiguous()
        assert S == plan.S
        E = plan.E
        T, nb, Tp = geo.block, geo.n_blocks, geo.score_pitch
        NPP = geo.n_padded                                             # (= NP unless the last block is ragged)
        if not (0 < NP <= NPP and NPP - NP < T and NP % 4 == 0):
            raise _lib.CsnError(f"{NP} points do not fill {nb} blocks of {T} (the last block may be short; counts are multiples of 4)")
        dev = x_all.device
        temperature = geo.temperature or float(d) ** 0.5               # csa_models.py:54
        w_qkv = torch.cat((w_qs, w_ks, w_vs), dim=0).contiguous()      # (3D, C)
        a16 = ctx.a16
        fwd16 = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}[a16]           # type of the forward's maps Qs, Ctx
        att = torch.empty((E, D, NP), device=dev, dtype=fwd16)
        lse = torch.empty((E, H, NPP), device=dev, dtype=torch.float32)
        # values may come from a different slot than the keys (slot kv + v_shift): only the generic
        # MultiHeadAttention.forward(Q, K, V) with three distinct inputs uses that
        tiles = fast_math() and tuning.current().kv_tiles and T <= 512
        if tiles:
            # 16-bit modes: K and V leave the projection as "tile planes" (per row and block 16 tiles of [hi 32 | lo 32] bf16 in
            # bf16x3, of [32] bf16 / fp16 in the single-product modes), which the attention kernels stage with plain copies; Q
            # (pre-scaled) stays fp32
            npl = planes()
            ldp = nb * 512 * npl
            qkv = torch.empty((S, D, NP), device=dev, dtype=fwd16)                            # Qs
            kv_dtype = torch.float16 if ctx.mode == 3 else torch.bfloat16
            # (the projection writes the zero padding of every block's last 32-key tile itself)
            kv = torch.empty((S, 2 * D, ldp), device=dev, dtype=kv_dtype)
            one_pass = plan.q_ranges is None and plan.kv_ranges is None and not a16 and tuning.current().qkv_one_pass
            if one_pass:
                # Q, K and V of every slot from ONE pass over x (csn_project_qkv_f32; the same bits as the two calls below)
                _lib.check(L.csn_project_qkv_f32(_ptr(x_all), C * NP, NP, _ptr(w_qkv), D, C, _ptr(qkv), D * NP, NP, _ptr(kv),
                                                 2 * D * ldp, ldp, S, NP, temperature, T, _stream()), "csn_project_qkv_f32")
            for first, step, count in ([] if one_pass else (plan.q_ranges or [(0, 1, S)])):
                _lib.check(L.csn_project_f32(x_all.data_ptr() + 4 * first * C * NP, step * C * NP, NP, _ptr(w_qkv), D, C,
                                             qkv.data_ptr() + qkv.element_size() * first * D * NP, step * D * NP, NP, count, NP, D,
                                             temperature, 3 if a16 else 0, 0, _stream()), "csn_project_f32")
            for first, step, count in ([] if one_pass else (plan.kv_ranges or [(0, 1, S)])):
                _lib.check(L.csn_project_f32(x_all.data_ptr() + 4 * first * C * NP, step * C * NP, NP, _ptr(w_qkv[D:]), 2 * D, C,
                                             kv.data_ptr() + 2 * first * 2 * D * ldp, step * 2 * D * ldp, ldp, count, NP, 0, 1.0, 2,
                                             T, _stream()), "csn_project_f32")
            q_ptr, q_stride = qkv.data_ptr(), D * NP
            k_ptr, kv_stride = kv.data_ptr(), 2 * D * ldp
            v_ptr = k_ptr + 2 * (D * ldp + v_shift * kv_stride)
            kv_flag, kv_pitch = 1, ldp
        else:
            kv = None
            if plan.q_ranges is None:
                qkv = project(x_all, w_qkv, div_rows=D, temperature=temperature)              # (S, 3D, NP); Q rows pre-scaled
            else:
                qkv = torch.empty((S, 3 * D, NP), device=dev, dtype=torch.float32)
                for rows0, nrows, ranges in ((0, D, plan.q_ranges), (D, 2 * D, plan.kv_ranges)):
                    for first, step, count in ranges:
                        _lib.check(L.csn_project_f32(x_all.data_ptr() + 4 * first * C * NP, step * C * NP, NP, _ptr(w_qkv[rows0:]),
                                                     nrows, C, qkv.data_ptr() + 4 * (first * 3 * D + rows0) * NP, step * 3 * D * NP, NP,
                                                     count, NP, D if rows0 == 0 else 0, temperature, 0, 0, _stream()),
                                   "csn_project_f32")
            q_ptr, q_stride = qkv.data_ptr(), 3 * D * NP
            k_ptr, kv_stride = q_ptr + 4 * D * NP, 3 * D * NP
            v_ptr = q_ptr + 4 * (2 * D * NP + v_shift * kv_stride)
            kv_flag, kv_pitch = 0, 0
        # attention backward data flow: keep the raw scores for it, or only lse (the backward then rebuilds S from Qs and K)
        flow = _score_flow(ctx.mode, d, T) if (keep_scores and tiles) else tuning.KEEP_SCORES
        scores = torch.empty((E, H, nb, T, Tp), device=dev, dtype=torch.float32) if (keep_scores and flow == tuning.KEEP_SCORES) else None
        # score storage: tile-major where the whole chain (forward, grouped dQ call, grouped dK / dV products on the 256 x 256
        # tiles) takes it — bf16x3 at d = 256 with kept scores; nothing outside this function reads the buffer
        tune = tuning.current()
        ctx.sc_layout = 1 if (scores is not None and tiles and ctx.mode == 1 and tune.tile_major_scores and tune.grouped_dq and
                              tune.grouped_dkv and Tp >= (T + 31) // 32 * 32 and (L.csn_attn_bwd_grouping(d, T) & 19) == 19) else 0
        with score_layout(ctx.sc_layout):
            if tune.grouped_fwd and fast_math() and plan.n_q_groups < E:
                # evaluations that share their query slot (the query shape's K+2) run in one work-group, Qs staged once
                _lib.check(L.csn_block_attn_fwd_grouped_f32(q_ptr, k_ptr, v_ptr, q_stride, kv_stride,
                                                            _ptr(q_slots), _ptr(kv_slots), NP, _ptr(att), D * NP, _ptr(scores),
                                                            _ptr(lse), E, H, d, T, nb, Tp, RESCALE_THRESHOLD, p_attn, seed_attn,
                                                            kv_flag, kv_pitch, _ptr(plan.q_group_items), _ptr(plan.q_group_off),
                                                            plan.n_q_groups, _stream()),
                           "csn_block_attn_fwd_grouped_f32")
            else:
                _lib.check(L.csn_block_attn_fwd_f32(q_ptr, k_ptr, v_ptr, q_stride, kv_stride,
                                                    _ptr(q_slots), _ptr(kv_slots), NP, _ptr(att), D * NP, _ptr(scores),
                                                    _ptr(lse), E, H, d, T, nb, Tp, RESCALE_THRESHOLD, p_attn, seed_attn,
                                                    kv_flag, kv_pitch, _stream()),
                           "csn_block_attn_fwd_f32")
        xhat = torch.empty((E, C, NP), device=dev, dtype=torch.float16 if a16 else torch.float32)
        rstd = torch.empty((E, NP), device=dev, dtype=torch.float32)
        w_fc = w_fc.contiguous()
        # third output (want_sums): sums[e][c] = sum_n xhat[e][c][n], formed in the epilogue of the out-projection (per-tile
        # partials in sum_ws) instead of a separate streaming pass over the 2.6 GB of maps
        sums = torch.empty((E, C), device=dev, dtype=torch.float32) if want_sums else None
        sum_ws_n = int(L.csn_outproj_ln_workspace_floats(E, C, D, NP)) if (want_sums and tuning.current().fused_point_sums) else 0
        sum_ws = torch.empty((sum_ws_n,), device=dev, dtype=torch.float32) if sum_ws_n else None
        _lib.check(L.csn_outproj_ln_fwd_f32(_ptr(att), D * NP, _ptr(w_fc), _ptr(x_all), C * NP, _ptr(q_slots),
                                            _ptr(xhat), C * NP, _ptr(rstd), E, C, D, NP, NP, LN_EPS, p_fc, seed_fc,
                                            _ptr(sums), _ptr(sum_ws), sum_ws_n, _stream()),
                   "csn_outproj_ln_fwd_f32")
        del sum_ws
        if keep_scores:
            ctx.save_for_backward(x_all, w_qkv, w_fc, qkv, att, lse, scores, xhat, rstd, kv)
            ctx.flow = flow
            ctx.geo = geo
            ctx.plan = plan
            ctx.drop = (p_attn, seed_attn, p_fc, seed_fc)
            ctx.ptrs = (q_stride, kv_stride, kv_flag, kv_pitch)
        # second output: the first n_head_evals maps again (same storage).  A consumer of all maps whose gradient is
        # constant along the points (the pooled means) and a consumer of the leading maps only (the mix) then hand the
        # backward two cheap gradients instead of one dense (E, C, NP) sum.
        ctx.n_head = n_head_evals
        # fourth output: a one-element handle.  A linked mix (csa_mix on LinkedMaps) takes IT as its differentiable input
        # and leaves the gradient of the mixed features in ctx.link instead of returning per-evaluation gradient maps
        handle = x_all.new_zeros(1) if link is not None else None
        if a16:
            # fp16 maps are data for the linked mix and nothing else: gradients reach this function through the link, the sums
            # and the handle, never through the maps
            head = xhat[:n_head_evals]
            ctx.mark_non_differentiable(xhat, head)
            return xhat, head, sums, handle
        return xhat, xhat[:n_head_evals], sums, handle

    @staticmethod
    def backward(ctx, dxhat, dhead, dsums=None, dhandle=None):
        with math_mode(backward_mode(ctx.mode)):
            # 16-bit maps: the gradient maps dQ / dK / dV too, where every slot is written once (grouped calls, flash)
            g16 = 0
            if ctx.a16:
                geo, tune = ctx.geo, ctx.tune
                grouping = _lib.lib().csn_attn_bwd_grouping(geo.d_head, geo.block)
                dq_once = ctx.flow != tuning.KEEP_SCORES or (tune.grouped_dq and (grouping & 1))
                dkv_once = ctx.flow == tuning.FLASH or (tune.grouped_dkv and (grouping & 2))
                g16 = 4 if (dq_once and dkv_once) else 0
            ctx.g16 = g16
            with act16(ctx.a16 + g16), score_layout(getattr(ctx, "sc_layout", 0)):
                return _MHAEvals._backward(ctx, dxhat, dhead, dsums, dhandle)

    @staticmethod
    def _backward(ctx, dxhat, dhead, dsums=None, dhandle=None):
        x_all, w_qkv, w_fc, qkv, att, lse, scores, xhat, rstd, kv = ctx.saved_tensors
        geo: MHAGeometry = ctx.geo
        plan: EvalPlan = ctx.plan
        L = _lib.lib()
        S, C, NP = x_all.shape
        H, d, D = geo.n_head, geo.d_head, geo.d_inner
        E = plan.E
        T, nb, Tp = geo.block, geo.n_blocks, geo.score_pitch
        dev = x_all.device
        # incoming gradient = dense maps for the first n_dense evaluations + a per-row constant (gradient of a mean)
        rows = None
        if dxhat is not None and dxhat.stride(2) == 0 and dxhat.stride(1) == 1:
            rows, dxhat = dxhat[:, :, 0].contiguous(), None              # expanded (E, C, 1) -> (E, C)
        if dsums is not None:                                            # gradient of the fused point sums: the same row term
            rows = dsums.contiguous() if rows is None else rows + dsums
        if ctx.n_head == 0:
            dhead = None
        if dxhat is not None and dhead is not None:
            dxhat = dxhat.clone()
            dxhat[:ctx.n_head] += dhead
            dhead = None
        dense = dxhat if dxhat is not None else dhead
        n_dense = 0 if dense is None else dense.shape[0]
        dense = None if dense is None else dense.contiguous()
        # linked mix: the dense term of the leading maps is scale[e][c] * dfeats[e // group] — rebuilt inside the kernel
        scale, group = None, 1
        link, ctx.link = ctx.link, None
        if dhandle is not None and link is not None and link.dfeats is not None:
            if dense is None:
                dense, scale, group, n_dense = link.dfeats, link.scale, link.group, ctx.n_head
            else:                                                        # somebody also used the maps densely: materialise
                dense = dense.clone() if dense.shape[0] == E else torch.cat(
                    (dense, dense.new_zeros((E - dense.shape[0],) + tuple(dense.shape[1:]))), dim=0)
                dense[:ctx.n_head] += link.scale[:, :, None] * link.dfeats.repeat_interleave(link.group, dim=0)[:ctx.n_head]
                n_dense = E
            link.dfeats = link.scale = None
        temperature = geo.temperature or float(d) ** 0.5
        p_attn, seed_attn, p_fc, seed_fc = ctx.drop
        need_dx = ctx.needs_input_grad[0]

        # ---- LayerNorm + fc backward -------------------------------------------------------------------
        a16 = ctx.a16
        bwd16 = torch.bfloat16 if a16 else torch.float32               # type of the backward's maps dZ, dCtx
        dz = torch.empty((E, C, NP), device=dev, dtype=bwd16)
        dz_res = torch.empty((E, C, NP), device=dev, dtype=torch.float32) if (need_dx and (p_fc > 0 or a16)) else None
        datt = torch.empty((E, D, NP), device=dev, dtype=bwd16)
        dw_fc = torch.empty((C, D), device=dev, dtype=torch.float32)
        ws_n = L.csn_wgrad_workspace_floats(C, D, E, NP)
        ws = torch.empty((ws_n,), device=dev, dtype=torch.float32)
        w_fc_t = w_fc.t().contiguous()
        _lib.check(L.csn_outproj_ln_bwd_f32(_ptr(dense), _ptr(xhat), _ptr(rstd), C * NP, _ptr(att), D * NP,
                                            _ptr(w_fc_t), _ptr(dz), _ptr(dz_res), _ptr(datt), _ptr(dw_fc), _ptr(ws), ws_n,
                                            E, C, D, NP, NP, 0, p_fc, seed_fc, 0, 0, _ptr(rows), n_dense, _ptr(scale), group,
                                            _stream()),
                   "csn_outproj_ln_bwd_f32")
        del ws

        # ---- attention backward, straight into per-slot gradient maps ---------------------------------------
        # evaluations that share a slot (Q of the query shape, K/V of each neighbour) add up: one launch per colour
        flow = ctx.flow
        pt = 1 if (fast_math() and Tp >= (T + 31) // 32 * 32) else 0    # P / dS travel to the dV / dK products as tile planes
        # (bf16x3: P overwrites the scores in place, dS fills `dscores`; bf16: both go to `dscores` as compact rows and the
        #  dK / dV call reads them there — csn_hip.h (3))
        if flow == tuning.KEEP_SCORES:
            dscores = torch.empty_like(scores)
        else:
            # recomputed scores: `scores` is only the scratch the P planes travel in (two-plane mode), `dscores` that of dS
            # (+ P in the one-plane mode); the flash flow has neither
            assert pt == 1 and scores is None
            shape = (E, H, nb, T, Tp)
            travel = flow == tuning.RECOMPUTE_DQ
            scores = torch.empty(shape, device=dev, dtype=torch.float32) if (travel and planes() == 2) else None
            dscores = torch.empty(shape, device=dev, dtype=torch.float32) if travel else None
        delta = torch.empty((E, H, geo.n_padded), device=dev, dtype=torch.float32)
        dqkv = torch.empty((S, 3 * D, NP), device=dev, dtype=torch.bfloat16 if ctx.g16 else torch.float32)
        ges = dqkv.element_size()
        # the weight gradients contract every slot's gradient maps — or, for a plan with slot ranges (and no input gradients
        # wanted), only the ranges: the maps of the other (slot, projection) pairs are then neither cleared nor read
        ranged = plan.q_ranges is not None and not need_dx
        if not ranged:
            if plan.q_unwritten.numel():
                dqkv[:, :D].index_fill_(0, plan.q_unwritten, 0.0)
            if plan.k_unwritten.numel():
                dqkv[:, D:2 * D].index_fill_(0, plan.k_unwritten, 0.0)
            if plan.v_unwritten.numel():
                dqkv[:, 2 * D:].index_fill_(0, plan.v_unwritten, 0.0)
        slot_stride = 3 * D * NP                                   # of the gradient maps, in elements
        q_stride, kv_stride, kv_flag, kv_pitch = ctx.ptrs
        gbase, q_ptr = dqkv.data_ptr(), qkv.data_ptr()
        # fp16 forward / bf16 backward: the forward's K / V planes hold fp16 bits; the attention backward kernels convert every
        # piece to bf16 in registers while they stage it (no second projection of K and V)
        kv_f16 = 1 if (kv_flag and ctx.mode == 3) else 0
        if kv_flag:
            k_ptr = kv.data_ptr()
            v_ptr = k_ptr + 2 * (D * kv_pitch + plan.v_shift * kv_stride)
        else:
            k_ptr = q_ptr + 4 * D * NP
            v_ptr = q_ptr + 4 * (2 * D * NP + plan.v_shift * kv_stride)
        grouping = L.csn_attn_bwd_grouping(d, T)
        tune = ctx.tune                                              # the forward's snapshot
        if flow != tuning.KEEP_SCORES:
            # one grouped call, scores rebuilt from the pre-scaled queries of the evaluation's query slot
            _lib.check(L.csn_block_attn_bwd_dq_recompute_f32(_ptr(datt), _ptr(att), D * NP, q_ptr, q_stride, _ptr(plan.q_slots),
                                                             k_ptr, v_ptr, kv_stride, _ptr(plan.kv_slots), NP, _ptr(scores),
                                                             _ptr(dscores), _ptr(lse), _ptr(delta), gbase, slot_stride,
                                                             _ptr(plan.q_slots), 0, _ptr(plan.q_group_items), E, H, d, T, nb, Tp,
                                                             p_attn, seed_attn, kv_pitch, kv_f16, pt if flow == tuning.RECOMPUTE_DQ else 0,
                                                             _ptr(plan.q_group_off), plan.n_q_groups, _stream()),
                       "csn_block_attn_bwd_dq_recompute_f32")
        elif tune.grouped_dq and (grouping & 1):
            # one call: the evaluations of a query slot run one after the other into the same dQ accumulators
            _lib.check(L.csn_block_attn_bwd_dq_f32(_ptr(datt), _ptr(att), D * NP, k_ptr, v_ptr, kv_stride,
                                                   _ptr(plan.kv_slots), NP, _ptr(scores), _ptr(dscores), _ptr(lse),
                                                   _ptr(delta), gbase, slot_stride, _ptr(plan.q_slots), 0,
                                                   _ptr(plan.q_group_items), E, H, d, T, nb, Tp, p_attn, seed_attn, 0, 0,
                                                   kv_flag + kv_f16, kv_pitch, pt, _ptr(plan.q_group_off), plan.n_q_groups, _stream()),
                       "csn_block_attn_bwd_dq_f32")
        else:
            for ci, ids in enumerate(plan.dq_colors):
                _lib.check(L.csn_block_attn_bwd_dq_f32(_ptr(datt), _ptr(att), D * NP, k_ptr, v_ptr, kv_stride,
                                                       _ptr(plan.kv_slots), NP, _ptr(scores), _ptr(dscores), _ptr(lse),
                                                       _ptr(delta), gbase, slot_stride, _ptr(plan.q_slots),
                                                       0 if ci == 0 else 1, _ptr(ids),
                                                       ids.numel(), H, d, T, nb, Tp, p_attn, seed_attn, 0, 0, kv_flag + kv_f16,
                                                       kv_pitch, pt, None, 0, _stream()),
                           "csn_block_attn_bwd_dq_f32")
        if flow == tuning.FLASH:
            # key-stationary kernel: P and dS are rebuilt per (key/value slot, head, block, 128 keys) from lse, delta and the
            # masks' seed; the evaluations of a key/value slot accumulate in registers (grouped) — no score-sized tensor exists
            _lib.check(L.csn_block_attn_bwd_dkv_flash_f32(_ptr(datt), D * NP, q_ptr, q_stride, _ptr(plan.q_slots), k_ptr, v_ptr,
                                                          kv_stride, _ptr(plan.kv_slots), kv_pitch, kv_f16, NP, _ptr(lse), _ptr(delta),
                                                          gbase + ges * D * NP, gbase + 2 * ges * D * NP, slot_stride,
                                                          _ptr(plan.kv_slots), _ptr(plan.v_slots), 0, _ptr(plan.kv_group_items),
                                                          E, H, d, T, nb, Tp, p_attn, seed_attn, _ptr(plan.kv_group_off),
                                                          plan.n_kv_groups, _stream()), "csn_block_attn_bwd_dkv_flash_f32")
        elif tune.grouped_dkv and (grouping & 2):
            # one call: the evaluations of a key/value slot are contracted one after the other into the same accumulators
            _lib.check(L.csn_block_attn_bwd_dkv_f32(_ptr(datt), D * NP, q_ptr, q_stride, _ptr(plan.q_slots), NP,
                                                    _ptr(scores), _ptr(dscores), gbase + ges * D * NP, gbase + 2 * ges * D * NP,
                                                    slot_stride, _ptr(plan.kv_slots), _ptr(plan.v_slots), 0,
                                                    _ptr(plan.kv_group_items), E, H, d, T, nb, Tp, 0, 0, 0, 0, pt,
                                                    _ptr(plan.kv_group_off), plan.n_kv_groups, _stream()),
                       "csn_block_attn_bwd_dkv_f32")
        else:
            for ci, ids in enumerate(plan.dkv_colors):
                _lib.check(L.csn_block_attn_bwd_dkv_f32(_ptr(datt), D * NP, q_ptr, q_stride, _ptr(plan.q_slots), NP,
                                                        _ptr(scores), _ptr(dscores), gbase + ges * D * NP, gbase + 2 * ges * D * NP,
                                                        slot_stride, _ptr(plan.kv_slots), _ptr(plan.v_slots),
                                                        0 if ci == 0 else 1, _ptr(ids),
                                                        ids.numel(), H, d, T, nb, Tp, 0, 0, 0, 0, pt, None, 0, _stream()),
                           "csn_block_attn_bwd_dkv_f32")
        del dscores, delta, datt

        # ---- projection weight gradients ------------------------------------------------------------------
        if ranged:
            dw_qkv = torch.empty((3 * D, C), device=dev, dtype=torch.float32)
            for rows0, nrows, ranges in ((0, D, plan.q_ranges), (D, 2 * D, plan.kv_ranges)):
                for i, (first, step, count) in enumerate(ranges):
                    ws_n = L.csn_wgrad_workspace_floats(nrows, C, count, NP)
                    ws = torch.empty((ws_n,), device=dev, dtype=torch.float32)
                    _lib.check(L.csn_project_wgrad_f32(dqkv.data_ptr() + ges * (first * 3 * D + rows0) * NP, step * 3 * D * NP, NP,
                                                       x_all.data_ptr() + 4 * first * C * NP, step * C * NP, NP,
                                                       dw_qkv.data_ptr() + 4 * rows0 * C, nrows, C, count, NP, 1.0,
                                                       0 if i == 0 else 1, _ptr(ws), ws_n, _stream()), "csn_project_wgrad_f32")
        else:
            dw_qkv = project_wgrad(dqkv, x_all)
        dw_q = dw_qkv[:D] / temperature               # Qs = (x Wq^T) / sqrt(d)  (csa_models.py:139)
        dw_k, dw_v = dw_qkv[D:2 * D], dw_qkv[2 * D:]

        dx_all = None
        if need_dx:
            # residual path + the three projections (not needed by the reference's training: inputs are constants)
            dqkv[:, :D] /= temperature
            dx_all = project(dqkv, w_qkv.t().contiguous())                   # (bf16 gradient maps: read as such)
            dx_all.index_add_(0, plan.q_slots.long(), dz if dz_res is None else dz_res)
        return dx_all, dw_q, dw_k, dw_v, dw_fc, None, None, None, None, None, None, None, None


class MixLink:
    """What a linked mix leaves for the backward of the evaluations it consumed: the gradient of the mixed features
    (one map per query shape), the per-(evaluation, channel) factors comp * gamma, and the evaluations mixed per shape."""
    __slots__ = ("dfeats", "scale", "group")

    def __init__(self):
        self.dfeats = self.scale = None
        self.group = 1


class LinkedMaps:
    """The leading maps of an evaluation batch as csa_mix wants them: the data (detached), the handle through which the
    gradient connection runs, and the link the two backward passes share."""
    __slots__ = ("maps", "handle", "link")

    def __init__(self, maps, handle, link):
        self.maps, self.handle, self.link = maps, handle, link


def mha_evals(x_all: torch.Tensor, w_qs: torch.Tensor, w_ks: torch.Tensor, w_vs: torch.Tensor, w_fc: torch.Tensor,
              plan: EvalPlan, geo: MHAGeometry, p_attn: float = 0.0, p_fc: float = 0.0, n_head_evals: int = 0,
              want_sums: bool = False, link_mix: bool = False):
    """p_attn / p_fc: train-mode dropout probabilities of csa_models.py:141 / :115 (0 in eval mode).
    n_head_evals > 0: returns (xhat, xhat[:n_head_evals]) — use the second for consumers of the leading maps only.
    want_sums: a further result, the (E, C) sums over the points of every map (differentiable; the pooled descriptors).
    link_mix: the second result is a LinkedMaps for csa_mix — the mix's backward then hands this function's backward the
    gradient of the mixed features and its factors instead of writing one gradient map per evaluation."""
    keep = torch.is_grad_enabled() and any(t.requires_grad for t in (x_all, w_qs, w_ks, w_vs, w_fc))
    link = MixLink() if (link_mix and tuning.current().link_mix and keep and n_head_evals > 0) else None
    xhat, head, sums, handle = _MHAEvals.apply(x_all, w_qs, w_ks, w_vs, w_fc, plan, geo, keep, float(p_attn), float(p_fc),
                                               int(n_head_evals), bool(want_sums), link)
    if link is not None:
        head = LinkedMaps(head.detach(), handle, link)
    out = (xhat, head) if n_head_evals > 0 else (xhat,)
    if want_sums:
        out = out + (sums,)
    return out if len(out) > 1 else out[0]


class _LinearCM(torch.autograd.Function):
    """y[s] = w @ x[s] on channel-major maps: x (S, C, NP), w (R, C) -> (S, R, NP).  R and C multiples of 4."""

    @staticmethod
    def forward(ctx, x, w):
        _need_cuda(x, w)
        ctx.save_for_backward(x, w)
        ctx.mode = current_mode()
        return project(x, w.contiguous())

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        with math_mode(backward_mode(ctx.mode)):
            dx = project(dy, w.t().contiguous()) if ctx.needs_input_grad[0] else None
            dw = project_wgrad(dy, x) if ctx.needs_input_grad[1] else None
        return dx, dw


def linear_cm(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """Bias-free 1x1 convolution on channel-major maps through the HIP GEMMs (the logit layer, csa_models.py:151)."""
    return _LinearCM.apply(x, w)


# ------------------------------------------------------------------------------------------------------
# pooled descriptors and the compatibility-weighted mix (csa_models.py:211-212, 218-219, 232-240)
# ------------------------------------------------------------------------------------------------------
class _RowSum(torch.autograd.Function):
    """(E, C, NP) -> (E, C): sum over the points of every channel row, accumulated in fp64 on the device."""

    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        E, C, NP = x.shape
        assert x.is_contiguous()
        out = torch.empty((E, C), device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().csn_rowsum_f32(_ptr(x), _ptr(out), E * C, NP, NP, _stream()), "csn_rowsum_f32")
        ctx.shape = (E, C, NP)
        return out

    @staticmethod
    def backward(ctx, g):
        return g[:, :, None].expand(ctx.shape)


def point_mean(x: torch.Tensor) -> torch.Tensor:
    """mean over points of channel-major maps, (E, C, NP) -> (E, C)  (csa_models.py:212, 219)."""
    return _RowSum.apply(x) / x.shape[-1]


class _CSAMix(torch.autograd.Function):
    """feats[b] = sum_k comp[b,k] * (gamma * xhat_k + beta).  xhat holds the maps [b*K1 + k]; when ``xself`` is given the
    k = 0 maps come from it ([b]) and xhat holds the K1 - 1 others ([b*(K1-1) + k-1])."""

    @staticmethod
    def forward(ctx, xhat, comp, gamma, beta, B: int, K1: int, xself=None):
        _need_cuda(xhat, comp, gamma, beta, xself)
        E, C, NP = xhat.shape
        assert xhat.is_contiguous() and E >= B * (K1 if xself is None else K1 - 1)
        assert xself is None or (xself.is_contiguous() and xself.shape == (B, C, NP))
        comp = comp.contiguous()
        feats = torch.empty((B, C, NP), device=xhat.device, dtype=torch.float32)
        _lib.check(_lib.lib().csn_mix_fwd_f32(_ptr(xhat), _ptr(comp), _ptr(gamma), _ptr(beta), _ptr(feats), B, K1, C, NP,
                                              _ptr(xself), _stream()), "csn_mix_fwd_f32")
        ctx.save_for_backward(xhat, comp, gamma, beta, xself)
        ctx.dims = (B, K1)
        return feats

    @staticmethod
    def backward(ctx, dfeats):
        xhat, comp, gamma, beta, xself = ctx.saved_tensors
        B, K1 = ctx.dims
        E, C, NP = xhat.shape
        dfeats = dfeats.contiguous()
        dxhat = torch.empty_like(xhat)
        n_mixed = B * (K1 if xself is None else K1 - 1)
        if E > n_mixed:
            dxhat[n_mixed:].zero_()                   # (callers hand in exactly the mixed maps: nothing to clear)
        dxself = torch.empty_like(xself) if xself is not None else None
        rowdot = torch.empty((B, K1, C), device=xhat.device, dtype=torch.float32)
        rowsum = torch.empty((B, C), device=xhat.device, dtype=torch.float32)
        _lib.check(_lib.lib().csn_mix_bwd_f32(_ptr(dfeats), _ptr(xhat), _ptr(comp), _ptr(gamma), _ptr(dxhat), _ptr(rowdot),
                                              _ptr(rowsum), B, K1, C, NP, _ptr(xself), _ptr(dxself), _stream()),
                   "csn_mix_bwd_f32")
        rd, rs = rowdot.double(), rowsum.double()
        g64, b64, c64 = gamma.double(), beta.double(), comp.double()
        dcomp = (rd * g64).sum(dim=2) + (rs * b64).sum(dim=1, keepdim=True)          # (B, K1)
        dgamma = torch.einsum("bk,bkc->c", c64, rd)
        dbeta = (c64.sum(dim=1, keepdim=True) * rs).sum(dim=0)
        return dxhat, dcomp.float(), dgamma.float(), dbeta.float(), None, None, dxself


class _CSAMixLinked(torch.autograd.Function):
    """_CSAMix on LinkedMaps: same forward; the backward computes only the reductions (d comp, d gamma, d beta) and leaves
    dfeats and comp * gamma in the links — csn_outproj_ln_bwd_f32 rebuilds comp_k gamma dfeats inside its LayerNorm backward,
    so the (B*K1, C, NP) gradient maps are neither written nor read."""

    @staticmethod
    def forward(ctx, handle, handle_self, comp, gamma, beta, B: int, K1: int, maps: LinkedMaps, maps_self):
        xhat = maps.maps
        xself = None if maps_self is None else maps_self.maps
        _need_f32(comp, gamma, beta)
        _need_cuda(xhat, xself, also=(torch.float16,))                               # (fp16 normalised maps of the 16-bit exchange)
        _need_cuda(comp, gamma, beta)
        E, C, NP = xhat.shape
        assert xhat.is_contiguous() and E == B * (K1 if xself is None else K1 - 1)
        assert xself is None or (xself.is_contiguous() and xself.shape == (B, C, NP))
        comp = comp.contiguous()
        feats = torch.empty((B, C, NP), device=xhat.device, dtype=torch.float32)
        x16 = xhat.dtype == torch.float16                                            # 16-bit activation maps (mha_evals)
        assert xself is None or (xself.dtype == xhat.dtype)
        with act16(1 if x16 else 0):
            _lib.check(_lib.lib().csn_mix_fwd_f32(_ptr(xhat), _ptr(comp), _ptr(gamma), _ptr(beta), _ptr(feats), B, K1, C, NP,
                                                  _ptr(xself), _stream()), "csn_mix_fwd_f32")
        ctx.save_for_backward(xhat, comp, gamma, beta, xself)
        ctx.dims = (B, K1)
        ctx.links = (maps.link, None if maps_self is None else maps_self.link)
        return feats

    @staticmethod
    def backward(ctx, dfeats):
        xhat, comp, gamma, beta, xself = ctx.saved_tensors
        B, K1 = ctx.dims
        E, C, NP = xhat.shape
        dfeats = dfeats.contiguous()
        rowdot = torch.empty((B, K1, C), device=xhat.device, dtype=torch.float32)
        rowsum = torch.empty((B, C), device=xhat.device, dtype=torch.float32)
        with act16(1 if xhat.dtype == torch.float16 else 0):
            _lib.check(_lib.lib().csn_mix_bwd_f32(_ptr(dfeats), _ptr(xhat), _ptr(comp), _ptr(gamma), None, _ptr(rowdot),
                                                  _ptr(rowsum), B, K1, C, NP, _ptr(xself), None, _stream()), "csn_mix_bwd_f32")
        factors = comp[:, :, None] * gamma                                           # (B, K1, C) = comp_k * gamma
        link, link_self = ctx.links
        if link_self is None:
            link.dfeats, link.scale, link.group = dfeats, factors.reshape(B * K1, C).contiguous(), K1
        else:
            link_self.dfeats, link_self.scale, link_self.group = dfeats, factors[:, 0].contiguous(), 1
            link.dfeats, link.scale, link.group = dfeats, factors[:, 1:].reshape(B * (K1 - 1), C).contiguous(), K1 - 1
        rd, rs = rowdot.double(), rowsum.double()
        g64, b64, c64 = gamma.double(), beta.double(), comp.double()
        dcomp = (rd * g64).sum(dim=2) + (rs * b64).sum(dim=1, keepdim=True)          # (B, K1)
        dgamma = torch.einsum("bk,bkc->c", c64, rd)
        dbeta = (c64.sum(dim=1, keepdim=True) * rs).sum(dim=0)
        one = dfeats.new_ones(1)                     # the handles' "gradient": its arrival is the signal, not its value
        return one, (one if link_self is not None else None), dcomp.float(), dgamma.float(), dbeta.float(), None, None, None, None


def csa_mix(xhat, comp: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, B: int, K1: int, xself=None) -> torch.Tensor:
    """sum_k comp[b,k] * (gamma * xhat[b,k] + beta) over channel-major maps.  xhat / xself: tensors, or the LinkedMaps that
    mha_evals(link_mix=True) returns (then no per-evaluation gradient maps exist in the backward)."""
    if isinstance(xhat, LinkedMaps):
        if xself is not None and not isinstance(xself, LinkedMaps):
            raise TypeError("csa_mix: xhat and xself must both be LinkedMaps or both tensors")
        return _CSAMixLinked.apply(xhat.handle, None if xself is None else xself.handle, comp, gamma, beta, B, K1, xhat, xself)
    return _CSAMix.apply(xhat, comp, gamma, beta, B, K1, xself)


# ------------------------------------------------------------------------------------------------------
# the compatibility head (csa_models.py:222-230), one launch forward and two backward instead of ~45 library launches
# ------------------------------------------------------------------------------------------------------
class _CompatHead(torch.autograd.Function):
    """comp (B, K+1) = softmax_k <normalize(Wq y_0 + bq), normalize(Wk key_k + bk)> over the pooled descriptors y (B, K+1, C)."""

    @staticmethod
    def forward(ctx, pooled, wq, bq, wk, bk, reference_layout: bool):
        _need_cuda(pooled, wq, bq, wk, bk)
        B, K1, C = pooled.shape
        pooled = pooled.contiguous()
        dev = pooled.device
        comp = torch.empty((B, K1), device=dev, dtype=torch.float32)
        save_u = torch.empty((B, K1 + 1, C), device=dev, dtype=torch.float64)
        save_n = torch.empty((B, K1 + 1), device=dev, dtype=torch.float64)
        wq_t, wk_t = wq.t().contiguous(), wk.t().contiguous()
        _lib.check(_lib.lib().csn_compat_fwd_f32(_ptr(pooled), _ptr(wq_t), _ptr(bq.contiguous()), _ptr(wk_t), _ptr(bk.contiguous()),
                                                 _ptr(comp), _ptr(save_u), _ptr(save_n), B, K1, C, 1 if reference_layout else 0,
                                                 _stream()), "csn_compat_fwd_f32")
        ctx.save_for_backward(pooled, wq, wk, comp, save_u, save_n)
        ctx.ref = reference_layout
        return comp

    @staticmethod
    def backward(ctx, dcomp):
        pooled, wq, wk, comp, save_u, save_n = ctx.saved_tensors
        B, K1, C = pooled.shape
        dev = pooled.device
        ws = torch.empty((2 * B * (K1 + 1) * C,), device=dev, dtype=torch.float64)
        dpooled = torch.empty_like(pooled)
        dwq, dwk = torch.empty_like(wq), torch.empty_like(wk)
        dbq, dbk = torch.empty((C,), device=dev, dtype=torch.float32), torch.empty((C,), device=dev, dtype=torch.float32)
        _lib.check(_lib.lib().csn_compat_bwd_f32(_ptr(dcomp.contiguous()), _ptr(comp), _ptr(save_u), _ptr(save_n), _ptr(pooled),
                                                 _ptr(wq.contiguous()), _ptr(wk.contiguous()), _ptr(ws), ws.numel(), _ptr(dpooled),
                                                 _ptr(dwq), _ptr(dbq), _ptr(dwk), _ptr(dbk), B, K1, C, 1 if ctx.ref else 0,
                                                 _stream()), "csn_compat_bwd_f32")
        return dpooled, dwq, dbq, dwk, dbk, None


def compat_head(pooled: torch.Tensor, wq: torch.Tensor, bq: torch.Tensor, wk: torch.Tensor, bk: torch.Tensor,
                reference_layout: bool = True) -> torch.Tensor:
    """The compatibility weights comp (B, K+1) of csa_models.py:222-230 from the pooled descriptors (B, K+1, C) and the two
    nn.Linear heads, with the reference's key-row bookkeeping for B > 1 (``reference_layout``) or per shape."""
    return _CompatHead.apply(pooled, wq, bq, wk, bk, bool(reference_layout))


# ------------------------------------------------------------------------------------------------------
# the loss the layers are trained with (csa_training.py:94-108)
# ------------------------------------------------------------------------------------------------------
class _MaskedCE(torch.autograd.Function):
    """logits (S, n_classes, N) class-major (any shape / class stride, points contiguous), labels (S, N) int64 ->
    stats (3,): mean cross-entropy over the points with mask < label < n_classes, their accuracy, their number."""

    @staticmethod
    def forward(ctx, logits, labels, mask):
        _need_cuda(logits)
        if labels.dtype != torch.int64 or not labels.is_cuda:
            raise _lib.CsnError(f"masked_cross_entropy takes int64 labels on the device (got {labels.dtype}, {labels.device})")
        S, n_cls, N = logits.shape
        if logits.stride(2) != 1 or tuple(labels.shape) != (S, N) or labels.stride(1) != 1:
            raise _lib.CsnError("masked_cross_entropy: logits (S, classes, N) with contiguous points, labels (S, N)")
        L = _lib.lib()
        dev = logits.device
        lse = torch.empty((S, N), device=dev, dtype=torch.float32)
        stats = torch.empty((3,), device=dev, dtype=torch.float32)
        ws_bytes = int(L.csn_masked_ce_workspace_bytes(S, N))
        ws = torch.empty((ws_bytes // 8,), device=dev, dtype=torch.float64)
        _lib.check(L.csn_masked_ce_fwd_f32(_ptr(logits), logits.stride(0), logits.stride(1), _ptr(labels), labels.stride(0), S, n_cls, N,
                                           int(mask), _ptr(lse), _ptr(ws), ws_bytes, _ptr(stats), _stream()), "csn_masked_ce_fwd_f32")
        ctx.save_for_backward(logits, labels, lse, stats)
        ctx.mask = int(mask)
        ctx.mark_non_differentiable(labels)
        return stats

    @staticmethod
    def backward(ctx, g):
        logits, labels, lse, stats = ctx.saved_tensors
        S, n_cls, N = logits.shape
        # only the loss (stats[0]) carries a gradient; accuracy and count are piecewise constant
        g0 = g[0:1].contiguous().float()
        dlogits = torch.empty((S, n_cls, N), device=logits.device, dtype=torch.float32)
        _lib.check(_lib.lib().csn_masked_ce_bwd_f32(_ptr(logits), logits.stride(0), logits.stride(1), _ptr(labels), labels.stride(0), S,
                                                    n_cls, N, ctx.mask, _ptr(lse), _ptr(stats), _ptr(g0), _ptr(dlogits), n_cls * N, N,
                                                    _stream()), "csn_masked_ce_bwd_f32")
        return dlogits, None, None


def masked_cross_entropy(logits: torch.Tensor, labels: torch.Tensor, mask: int = 0):
    """csa_training.py:94-108 on the device: (mean cross-entropy, accuracy, number of counted points) over the points whose label
    is above ``mask``; logits (S, n_classes, N[, 1]) class-major as the logit layer writes them, labels (S, N) int64.  One pass
    over the logits forward and one backward, instead of transpose + gather + log_softmax + nll of the reference's form."""
    if logits.dim() == 4:
        logits = logits.squeeze(-1)
    st = _MaskedCE.apply(logits, labels, mask)
    return st[0], st[1], st[2]
