"""Data-flow switches of the step and the bench's event hook — kept out of the product modules.

Every switch selects between two forms of the SAME arithmetic (a fused pass and the plain pass it replaced, or two data
flows of the attention backward); the defaults are the measured-faster forms.  Nothing here changes results
beyond fp32 rounding; `tests/test_gpu_module.py::test_fused_data_flow_equals_the_unfused_one` and
`tests/test_gpu_flash.py` flip them through ``override`` to check one form against the other (`bench.py` times the launches
through csn_amd._lib.set_call_hook, not through anything here).  The state is process-wide on purpose: autograd runs the backward on its own threads, and a step must see the
same switches in both passes.
"""
from __future__ import annotations

import contextlib
from dataclasses import dataclass, field, fields, replace
from typing import Dict, Optional

# attention backward data flows (csn_amd.functional._MHAEvals):
KEEP_SCORES = 0      # the forward writes the raw scores S, the dQ kernel reads them back and leaves P / dS for the dK / dV products
RECOMPUTE_DQ = 1     # the forward keeps only lse; the dQ kernel rebuilds S = Qs K^T (one more product), P / dS still travel
FLASH = 2            # ... and a key-stationary kernel rebuilds P / dS for dK / dV: no score-sized tensor exists at all


@dataclass
class Tuning:
    kv_tiles: bool = True            # 16-bit modes: K / V leave the projection as tile planes (csn_project_f32, out_split = 2)
    fused_point_sums: bool = True    # False: pooled sums by a streaming pass over the maps
    link_mix: bool = True            # False: the mix backward writes per-evaluation gradient maps
    grouped_dkv: bool = True         # False: dK / dV by one read-modify-write launch per colour
    grouped_dq: bool = True          # False: dQ likewise
    grouped_fwd: bool = True         # 16-bit modes: the forward's evaluations grouped by query slot (the query operand staged once per group)
    fused_compat_head: bool = True   # False: the compatibility head as torch ops (two nn.Linear, normalize, einsum, softmax)
    act16: bool = True               # bf16 / fp16 modes, linked mix: Qs, Ctx, xhat, dZ, dCtx between the launches as 16-bit maps
    # bf16x3, kept scores: S and the P / dS planes stored [key tile][query][32 keys], so that every wave instruction that touches
    # them moves 1 KB in one piece.  Measured +-0 over the config-3 step (27.69 vs 27.68 ms, profiles/r4k_ab_score_layout.txt):
    # the 16-byte pieces 2 KB apart were not what the dQ kernel or the dV / dK products wait for.  Off; kept as the measured
    # form (tests/test_gpu_score_layout.py holds it to the row-major step bit for bit)
    tile_major_scores: bool = False
    # Q, K and V of all slots from ONE pass over x (csn_project_qkv_f32: three row sets of the streaming kernel walking the same
    # chunks) where every slot needs all three.  The same bits, 1.15 GB less HBM traffic per step (FETCH / WRITE passes), and not
    # faster: the launch alone 1.96 ms against 0.54 + 1.06 + launch gap = 1.78 for the two calls, the config-3 step 26.95 against
    # 26.99 ms (profiles/r4t_qkv_one_pass.txt).  Off; kept as the measured form (tests/test_gpu_wx.py holds it bit for bit)
    qkv_one_pass: bool = False
    # attention backward data flow by (math mode of the backward: 1 bf16x3, 2 bf16 — fp16 forwards run their backward in 2;
    # head width), or by mode alone; taken where the kernels have an instance for it (csn_attn_bwd_grouping bits 2 / 3),
    # KEEP_SCORES otherwise.  Measured per mode and width, DESIGN.md §4 "data flow A/B": at d = 256 the extra matrix products
    # cost what the score traffic saves, at d <= 128 a score costs the same bytes for a fraction of the FLOPs
    score_flow: Dict[object, int] = field(default_factory=lambda: {
        1: KEEP_SCORES,                                   # bf16x3 at d = 256: three LDS images of two planes do not fit one CU
        2: RECOMPUTE_DQ,                                  # one plane, d = 256: -4.7 % of the config-3 step (profiles/r3l_flow_ab_step.txt)
        **{(2, d): FLASH for d in (32, 64, 96, 128)},     # config 5: 18.5 -> 17.1 (recompute) -> 15.4 ms (flash)
        **{(1, d): FLASH for d in (32, 64, 96)}, (1, 128): RECOMPUTE_DQ})

    def flow_for(self, mode: int, d_head: int) -> int:
        return self.score_flow.get((mode, d_head), self.score_flow.get(mode, KEEP_SCORES))


_current = Tuning()


def current() -> Tuning:
    return _current


@contextlib.contextmanager
def override(**changes):
    """``with tuning.override(grouped_dq=False): ...`` — the switches inside the block, the previous ones after it.
    NOT thread-safe: the object is process-wide and swapped without a lock.  A step is safe against it all the same — every
    forward snapshots the switches into its autograd context and its backward reads only that snapshot."""
    global _current
    names = {f.name for f in fields(Tuning)}
    unknown = set(changes) - names
    if unknown:
        raise TypeError(f"unknown tuning switch(es): {sorted(unknown)}")
    saved = _current
    _current = replace(saved, **changes)
    try:
        yield _current
    finally:
        _current = saved
