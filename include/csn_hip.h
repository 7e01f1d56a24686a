/* csn_hip.h — C ABI of libcsn_hip.so: the MI355X (gfx950) implementation of CSN's cross-shape-attention
 * hot path.  Plain pointers and sizes only; every pointer is a DEVICE pointer unless stated otherwise;
 * `stream` is a hipStream_t passed as void*.  All functions return 0 on success, a negative CSN_E_* code
 * on a rejected argument, or a positive hipError_t from the launch.  Nothing here allocates, synchronises
 * or touches the host: every entry point only enqueues kernels on `stream` (graph-capture safe).
 *
 * The reference (marios2019/CSN) has no native layer at all: its hot path is the PyTorch op sequence in
 * MID-FC/csa_models.py.  Each entry point below names the reference lines whose arithmetic it replaces;
 * the Python binding that a maintainer of the reference would add is shown in INTEGRATION.md and lives
 * in csn_amd/_lib.py.
 *
 * DATA LAYOUT.  Every activation is CHANNEL-MAJOR fp32: a per-shape feature map is [channels][points]
 * with leading dimension `ld` (>= points) — exactly the reference's (B, C, N, 1) input
 * (MID-FC/features_data_loader.py:45-48, csa_models.py:88-94).  "shape slot" = one 3D shape's feature map;
 * "evaluation" = one multi-head-attention call MHA(x_q; x_kv) of csa_models.py:81-125 (a CSA forward of
 * one query shape with K neighbours is 2K+1 distinct evaluations, csa_models.py:209-242).
 * Points are processed in `n_blocks` consecutive blocks of `block` points; query block i attends to
 * key/value block i only (csa_models.py:83-90: 20 blocks of 500).
 * Alignment contract (checked, CSN_E_ALIGN): device pointers 16-byte aligned; ld, block, n_points,
 * channel counts and all strides multiples of 4 floats; head dim in {32,64,96,128,256};
 * d_model in {32,64,96,128,256}.
 */
#ifndef CSN_HIP_H
#define CSN_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define CSN_ABI_VERSION 17
/* The library is built with -fvisibility=hidden: the entry points declared here are its ONLY exported symbols. */
#define CSN_API __attribute__((visibility("default")))

/* math modes (csn_set_math_mode / csn_set_thread_math_mode) */
#define CSN_MATH_FP32 0
#define CSN_MATH_BF16X3 1
#define CSN_MATH_BF16 2
#define CSN_MATH_FP16 3

#define CSN_E_ARG (-1)     /* null pointer / non-positive size / a count beyond its row (leading dimension) / bad flag */
#define CSN_E_ALIGN (-2)   /* a size or leading dimension is not % 4      */
#define CSN_E_PTR (-3)     /* pointer not 16-byte aligned                  */
#define CSN_E_STRIDE (-4)  /* a stride is not % 4                          */
#define CSN_E_DIM (-5)     /* unsupported head / model dimension           */
#define CSN_E_WORKSPACE (-6) /* workspace too small                        */

/* ABI version of the loaded library (== CSN_ABI_VERSION). */
CSN_API int csn_version(void);
/* Arithmetic of the contractions (projections, attention products, out-projection, every gradient product):
 *   0 CSN_MATH_FP32    exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32 / 16x16x4: bit-identical to an fmaf chain);
 *   1 CSN_MATH_BF16X3  every fp32 operand is split into two bf16 terms and a product is three bf16 MFMAs
 *                      (hi*hi + hi*lo + lo*hi, fp32 accumulate): ~1e-5 relative error per product, 5.3x the fp32 matrix rate.
 *                      THE DEFAULT: like mode 0 it is inside the 1e-4 contract of the path (every parity test runs in both);
 *   2 CSN_MATH_BF16    every operand rounded once to bf16, one MFMA per product, fp32 accumulate; tile planes hold one
 *                      plane (half the bytes).  OUTSIDE the 1e-4 contract: ~1e-2 relative on outputs (reported by the tests);
 *   3 CSN_MATH_FP16    the same with fp16 operands — FORWARD entry points only (gradients of this path reach 1e-7 and
 *                      underflow fp16): the backward entry points return CSN_E_ARG in this mode; callers run them in mode 2
 *                      (csn_amd does: "fp16 forward / bf16 backward").
 * The retrieval measure (7) always runs in exact fp32 (bit-exact kNN indices).  The cross-length entry points (3b) have no
 * single-product kernels: in modes 2 / 3 they run as mode 1.
 * csn_set_math_mode sets the PROCESS default; csn_set_thread_math_mode overrides it for the calling thread only (-1 clears the
 * override), which is how a module selects its own mode per call without touching other threads (csn_amd brackets every
 * forward / backward with it; autograd's backward threads set their own).  csn_get_math_mode returns the mode in effect for
 * the calling thread.  Both setters return CSN_E_ARG for an unknown value. */
/* SPLIT TENSORS (math mode 1).  Arguments named *_split / *_plane_stride let a kernel write, or read, an fp32
 * tensor as two bf16 planes x = hi + lo (hi = bf16(x), lo = bf16(x - hi)): the pointer then addresses the HIGH
 * plane (bf16 elements, cast to float* for the ABI), the LOW plane starts `plane_stride` bf16 elements later, and
 * every stride / leading dimension of that tensor counts bf16 elements (same numbers as for the fp32 tensor).
 * The producer's epilogue splits once; consumers stage the planes into LDS with plain copies (no conversion work per
 * tile).  A split dctx (attention-output gradient) is laid out [evaluation][2 planes][n_heads*d_head][ld]: its
 * evaluation stride is 2 * ctx_eval_stride and dctx_plane_stride = ctx_eval_stride.
 * TILE PLANES (math modes 1..3): the form in which keys and values travel from the projection to the attention kernels
 * (described for mode 1, two planes; in modes 2 / 3 there is ONE plane of bf16 / fp16 elements: a tile is [32 keys], the block
 * pitch 512, ld_out = n_blocks * 512, and a probs_tiles backward leaves P and dS as compact rows — see (3)).
 * csn_project_f32 with out_split = 2 writes, per output row and per attention block of `out_plane_stride` (<= 512) points,
 * 16 tiles of [hi: 32 keys | lo: 32 keys] bf16 — block pitch 1024, row pitch ld_out = n_blocks * 1024, out_shape_stride in
 * bf16 elements; the padding keys of a block's last 32-key tile are written as zeros (tiles beyond it are not touched and never read).  Every 32-key tile row is then
 * 128 contiguous, 128-byte aligned bytes that the attention kernels stage into LDS with plain copies.  The attention
 * entry points take such a tensor for k and v when qkv_split / kv_split != 0 (then *_plane_stride = that row pitch and
 * kv_shape_stride counts bf16 elements); q stays fp32.
 * STATUS: live are csn_project_f32 out_split (1: whole planes, 2: tile planes), csn_outproj_ln_bwd_f32 dctx_split, and the
 * k/v tile planes of csn_block_attn_fwd_f32 / csn_block_attn_bwd_dq_f32; dctx_split / q_split INPUTS are reserved
 * (CSN_E_ARG).  Passing *_split != 0 in math mode 0 returns CSN_E_ARG. */
/* 16-BIT ACTIVATION MAPS (math modes 2 and 3).  In the single-product modes every matrix operand is rounded to 16 bits when it
 * is staged, so the maps the entry points hand to each other can travel in that form: half the bytes, and the consumers stage
 * them by copy.  csn_set_thread_act16(fmt) switches the CALLING THREAD's calls to that exchange format — fmt 1: the forward's
 * maps are bf16 (a math-mode-2 forward), fmt 2: they are fp16 (a math-mode-3 forward; its backward runs in mode 2 and converts
 * them in registers while staging), 0: off (the default: everything below is fp32).  With the flag set
 *   Qs   (csn_project_f32 out_split = 3 writes it; read by (2), by the recomputing calls of (3) and by the dK product)   fwd type
 *   Ctx  (written by (2); read by (4) forward, by (3)'s delta and by the W_fc gradient of (4) backward)                 fwd type
 *   xhat (written by (4) forward; read by (4) backward and by (6))                                                      fp16 always (|xhat| < sqrt(C))
 *   dZ, dCtx (written by (4) backward; read by (3))                                                                     bf16
 *   dQ, dK, dV (fmt + 4 only: written by (3), read by csn_project_wgrad_f32 and, as x, by csn_project_f32 out_split + 16)    bf16
 * are ONE plane of 16-bit elements with the shapes documented below; their pointers stay typed float* and every stride and
 * leading dimension of such a map counts 16-bit ELEMENTS (same numbers as for the fp32 map).  Everything else keeps its type:
 * the input maps x, the residual, dxhat / dfeats, lse / delta / rstd, every gradient map dQ / dK / dV / dz_res, weights and
 * weight gradients, the K / V tile planes (already 16-bit).  Products are bit-identical to the fp32 exchange wherever a map is
 * only a matrix operand (it was rounded to the same 16 bits at staging); delta = sum dO * O, the LayerNorm backward and the mix
 * see the rounded maps (|xhat| error <= 2^-11 relative).  Requirements: math mode 2 / 3 with K / V tile planes, block mode (no
 * cross-length entry points), no split tensors, a mix backward without gradient maps (the linked form), evaluation outputs
 * written once (no accumulate into a 16-bit map).  A forward entry point returns CSN_E_ARG when fmt does not name its mode's
 * type; backward entry points take either fmt in mode 2.  fmt + 4 (5 or 6) also makes the gradient maps dQ / dK / dV bf16:
 * they are then written once per slot (grouped calls; accumulate != 0 returns CSN_E_ARG), and what contracts them rounds them to
 * bf16 anyway — the weight gradients are the same bits. */
CSN_API int csn_set_thread_act16(int fmt);
CSN_API int csn_get_thread_act16(void);
/* SCORE STORAGE of the block-attention entry points (3), (3c), per calling thread.  0 (default): the scores of a block are
 * [query][key] rows of pitch score_pitch — what a caller that reads probabilities expects.  1: TILE-MAJOR — per block
 * [key tile of 32][query][32 keys]: the forward's scores, and the P / dS tile planes the backward hands from its dQ call to its
 * dK / dV call, are then written and read in contiguous runs (a wave instruction moves 1 KB in one piece instead of sixteen
 * 64-byte pieces 2 KB apart).  Same arithmetic, same buffer sizes; only meaningful when the three calls of one evaluation batch
 * agree.  Taken where csn_attn_bwd_grouping reports bit 4 (bf16x3 mode, block mode, tile-plane K / V, probs_tiles = 1, the dK / dV
 * products on the 256 x 256 tiles); CSN_E_ARG otherwise.  csn_amd sets it around the training step's three calls. */
CSN_API int csn_set_thread_score_layout(int layout);
CSN_API int csn_get_thread_score_layout(void);
CSN_API int csn_set_math_mode(int mode);
CSN_API int csn_set_thread_math_mode(int mode);
CSN_API int csn_get_math_mode(void);
/* The calling thread's own override as set by csn_set_thread_math_mode (-1 = none): what a scoped override saves and restores. */
CSN_API int csn_get_thread_math_mode(void);
/* Human-readable text for a status code returned by any function below. Host pointer, static storage. */
CSN_API const char* csn_status_string(int status);

/* ---- (1) pre-attention projections ------------------------------------------------------------------
 * out[s][r][n] = sum_c w[r][c] * x[s][c][n],  r < rows, n < n_points; rows r < div_rows are then divided
 * by `temperature`.   Replaces w_qs / w_ks / w_vs (nn.Linear, no bias; csa_models.py:49-51,103-105) and the
 * `q / temperature` of csa_models.py:139 (stack W_q|W_k|W_v along rows and pass div_rows = n_head*d_k to
 * project a shape once for all of its evaluations).  out_split: 0 fp32 maps, 1 / 2 split tensors / tile planes (above), 3 (math
 * modes 2 / 3): ONE 16-bit map per shape in the mode's type — out_shape_stride and ld_out count 16-bit elements.  out_split + 16
 * (math mode 2): x itself is a bf16 map (x_shape_stride, ld_x in elements) — the input gradient W^T dqkv from bf16 gradient maps. */
CSN_API int csn_project_f32(const float* x, long long x_shape_stride, int ld_x, const float* w, int rows, int channels,
                    float* out, long long out_shape_stride, int ld_out, int n_shapes, int n_points, int div_rows,
                    float temperature, int out_split, long long out_plane_stride, void* stream);

/* Q, K and V of the same slots in one call (csa_models.py:103-105 on one input): w_qkv = [W_q ; W_k ; W_v] (3 d_inner rows),
 * Qs = W_q x / temperature as an fp32 map q_out [slot][d_inner][ld_q], K | V as tile planes kv_out [slot][2 d_inner][ld_kv] —
 * exactly what csn_project_f32(.., div_rows = d_inner, out_split = 0) and csn_project_f32(.., out_split = 2,
 * out_plane_stride = block) write, bit for bit; 16-bit math modes only (tile planes).  In the bf16x3 mode with 256 channels and
 * d_inner = 256 the three row sets share ONE pass over x (the streaming kernel); elsewhere it is the two calls. */
CSN_API int csn_project_qkv_f32(const float* x, long long x_shape_stride, int ld_x, const float* w_qkv, int d_inner, int channels,
                        float* q_out, long long q_shape_stride, int ld_q, void* kv_out, long long kv_shape_stride, int ld_kv,
                        int n_shapes, int n_points, float temperature, int block, void* stream);

/* ---- (2) block-diagonal scaled-dot-product attention, forward ----------------------------------------
 * For evaluation e, head h, block b:  P = softmax(Qs K^T) over the block's keys, ctx = P V
 * (ScaledDotProductAttention.forward, csa_models.py:138-144, eval mode; Qs is already divided by the
 * temperature by csn_project_f32).  q/k/v point at row 0 of the projected [n_heads*d_head][ld] maps of
 * slot 0; evaluation e reads slot q_index[e] (queries) and kv_index[e] (keys, values); NULL = identity.
 *   ctx    [n_evals][n_heads*d_head][ld]   (eval stride given)         — csa_models.py:114 before `fc`
 *   lse    [n_evals][n_heads][n_blocks*block]   log-sum-exp of every score row (for the backward); may be NULL (inference)
 *   scores [n_evals][n_heads][n_blocks][block][score_pitch]  raw scores S[query][key]; may be NULL
 *          (inference).  score_pitch >= block, % 4.
 * RAGGED LAST BLOCK: when n_blocks * block > ld the row of ld points ends inside the last block, which then holds
 * ld - (n_blocks - 1) * block points (a multiple of 4; CSN_E_ARG if that is not positive): queries and keys of the last block
 * are cut there in all three block-attention entry points.  lse / delta / scores keep their n_blocks * block layout.
 * rescale_threshold: the running softmax maximum is only re-based when it grows by more than this
 * (0 = re-base on every key tile); results agree to fp32 rounding for any value <= ~40.
 * dropout_p / seed: train-mode dropout on the probabilities (nn.Dropout(0.1), csa_models.py:133-141):
 * P_drop = mask * P / (1 - p) with a counter-based mask, a pure function of (seed, position in `scores`), so
 * the backward call regenerates it from the same (dropout_p, seed).  0 = eval mode. */
CSN_API int csn_block_attn_fwd_f32(const float* q, const float* k, const float* v, long long q_shape_stride,
                           long long kv_shape_stride, const int* q_index, const int* kv_index, int ld, float* ctx,
                           long long ctx_eval_stride, float* scores, float* lse, int n_evals, int n_heads,
                           int d_head, int block, int n_blocks, int score_pitch, float rescale_threshold,
                           float dropout_p, unsigned long long seed, int qkv_split, long long qkv_plane_stride,
                           void* stream);

/* The same forward with the evaluations GROUPED BY QUERY SLOT: group g = eval_ids[group_offsets[g] .. group_offsets[g+1])
 * (device int32 arrays; every evaluation of [0, n_evals) listed exactly once; all evaluations of a group have the same
 * q_index — CSN_E_ARG semantics are the caller's: the library does not read the arrays on the host).  In the 16-bit math modes a
 * work-group stages the pre-scaled queries of its 128-query tile ONCE and runs the group's evaluations one after the other
 * with the operand in registers (the query shape of a CSA step serves K+2 evaluations: csa_models.py:210, 232-237 — half of
 * the step's operand blocks are never fetched); in the fp32 mode the evaluations run ungrouped.  Every output is the
 * ungrouped call's, bit for bit. */
CSN_API int csn_block_attn_fwd_grouped_f32(const float* q, const float* k, const float* v, long long q_shape_stride,
                           long long kv_shape_stride, const int* q_index, const int* kv_index, int ld, float* ctx,
                           long long ctx_eval_stride, float* scores, float* lse, int n_evals, int n_heads,
                           int d_head, int block, int n_blocks, int score_pitch, float rescale_threshold,
                           float dropout_p, unsigned long long seed, int qkv_split, long long qkv_plane_stride,
                           const int* eval_ids, const int* group_offsets, int n_groups, void* stream);

/* ---- (3) block attention, backward (autograd of csa_models.py:139-142) -------------------------------
 * Two calls, because their outputs are shared differently between evaluations (the query shape's Q serves
 * K+1 evaluations, a neighbour's K/V serve two): each call processes the `n_launch_evals` evaluations listed in
 * `eval_ids` (NULL = 0..n-1) and writes into per-SLOT gradient maps, adding to them when `accumulate` != 0.  The
 * caller groups evaluations so that no two evaluations of one call share an output slot.
 *
 * csn_block_attn_bwd_dq_f32:  in  dctx, ctx [eval][n_heads*d_head][ld], k/v + kv_index as in forward,
 *                                 scores (S[query][key] from forward), lse;
 *                             out scores := P_drop (in place; = P without dropout), dscores := dS,
 *                                 delta [eval][n_heads][n_blocks*block] scratch = rowsum(dctx*ctx),
 *                                 dq[dq_index[e]] (+)= dS K   — gradient w.r.t. the pre-scaled queries Qs.
 * csn_block_attn_bwd_dkv_f32: in  dctx, q + q_index as in forward, probs (= scores after the dq call), dscores;
 *                             out dv[dv_index[e]] (+)= P^T dctx,  dk[dk_index[e]] (+)= dS^T Qs.
 * dq/dk/dv point at row 0 of the [n_heads*d_head][ld] gradient map of slot 0; *_slot_stride in floats.
 * probs_tiles != 0 (math modes 1, 2; score_pitch >= block rounded up to 32): the dq call leaves P_drop and dS as TILE
 * PLANES and the dkv call must be told the same; the dV / dK products then stage them with plain copies.  Mode 1: per query
 * row 16 tiles of [hi: 32 keys | lo: 32 keys] bf16 — the bytes of the fp32 row — P in place over `scores`, dS in `dscores`.
 * Mode 2 (one plane): rows of 16 tiles of [32 keys] are half the bytes, so both go to `dscores` — per (evaluation, head,
 * block) region of block * score_pitch floats: [P: block rows | dS: block rows] of pitch score_pitch bf16 elements — and
 * `scores` is left untouched; the dkv call reads both from its `dscores` argument (`probs` is ignored).  Mode 2 REQUIRES
 * kv_split and probs_tiles.
 * GROUPED calls (group_offsets != NULL): eval_ids lists the n_launch_evals evaluations ordered so that evaluations sharing
 * an output slot are adjacent, group g = entries group_offsets[g] .. group_offsets[g+1] (n_groups + 1 offsets); a group's
 * results are accumulated in registers and its slot is written once — one call for all evaluations instead of one
 * read-modify-write pass per colour.  csn_attn_bwd_grouping() says where that is available in the current math mode:
 * bit 0 = the dq call, bit 1 = the dkv call, bit 2 = csn_block_attn_bwd_dq_recompute_f32, bit 3 =
 * csn_block_attn_bwd_dkv_flash_f32 (both below), bit 4 = tile-major score storage (csn_set_thread_score_layout).
 *
 * csn_block_attn_bwd_dq_recompute_f32 — the dq call WITHOUT saved scores ("flash" data flow; math modes 1 and 2, K / V as tile
 * planes, block mode): the forward is run with scores = NULL (only lse is kept) and this call rebuilds S = Qs K^T tile by tile
 * from the pre-scaled queries q [slot][n_heads*d_head][ld] (+ q_index, as in the forward) — one more matrix product per tile
 * against 4 * block bytes of score traffic per query in each direction.  Everything else as csn_block_attn_bwd_dq_f32.
 *   probs_tiles != 0: P_drop and dS still leave as tile planes for csn_block_attn_bwd_dkv_f32 — mode 1: P in `probs`, dS in
 *                     `dscores` (two scratch buffers of the scores' geometry); mode 2: both in `dscores`, `probs` unused;
 *   probs_tiles == 0: nothing is written besides delta and dq (`probs` / `dscores` may be NULL) — for a dK / dV pass that
 *                     recomputes the probabilities itself.
 * Available where three LDS tile images per stage fit: mode 2 at every head width, mode 1 up to d_head = 128 (bit 2 of
 * csn_attn_bwd_grouping); CSN_E_ARG elsewhere.
 * kv_f16 != 0 (math mode 2 only; also csn_block_attn_bwd_dkv_flash_f32, and kv_split = 2 of csn_block_attn_bwd_dq_f32): k and
 * v are the tile planes of a forward that ran in math mode 3 — fp16 bits.  The kernels convert every piece to bf16 in
 * registers on its way into LDS, so the "fp16 forward / bf16 backward" pairing needs no second projection of K and V. */
CSN_API int csn_attn_bwd_grouping(int d_head, int block);
CSN_API int csn_block_attn_bwd_dq_recompute_f32(const float* dctx, const float* ctx, long long ctx_eval_stride, const float* q,
                                        long long q_shape_stride, const int* q_index, const float* k, const float* v,
                                        long long kv_shape_stride, const int* kv_index, int ld, float* probs,
                                        float* dscores, const float* lse, float* delta, float* dq,
                                        long long dq_slot_stride, const int* dq_index, int accumulate, const int* eval_ids,
                                        int n_launch_evals, int n_heads, int d_head, int block, int n_blocks,
                                        int score_pitch, float dropout_p, unsigned long long seed,
                                        long long kv_plane_stride, int kv_f16, int probs_tiles, const int* group_offsets,
                                        int n_groups, void* stream);
CSN_API int csn_block_attn_bwd_dq_f32(const float* dctx, const float* ctx, long long ctx_eval_stride, const float* k,
                              const float* v, long long kv_shape_stride, const int* kv_index, int ld, float* scores,
                              float* dscores, const float* lse, float* delta, float* dq, long long dq_slot_stride,
                              const int* dq_index, int accumulate, const int* eval_ids, int n_launch_evals, int n_heads,
                              int d_head, int block, int n_blocks, int score_pitch, float dropout_p,
                              unsigned long long seed, int dctx_split, long long dctx_plane_stride, int kv_split,
                              long long kv_plane_stride, int probs_tiles, const int* group_offsets, int n_groups,
                              void* stream);
CSN_API int csn_block_attn_bwd_dkv_f32(const float* dctx, long long ctx_eval_stride, const float* q, long long q_shape_stride,
                               const int* q_index, int ld, const float* probs, const float* dscores, float* dk,
                               float* dv, long long dkv_slot_stride, const int* dk_index, const int* dv_index,
                               int accumulate, const int* eval_ids, int n_launch_evals, int n_heads, int d_head,
                               int block, int n_blocks, int score_pitch, int dctx_split, long long dctx_plane_stride,
                               int q_split, long long q_plane_stride, int probs_tiles, const int* group_offsets,
                               int n_groups, void* stream);

/* csn_block_attn_bwd_dkv_flash_f32 — dK and dV WITHOUT P / dS tensors (bit 3 of csn_attn_bwd_grouping: math modes 1 and 2,
 * d_head <= 128, block mode): a work-group keeps 128 keys (256 at d_head = 96 in math mode 2) of one (key/value slot, head, block)
 * in registers, streams the
 * pre-scaled queries q and the output gradient dctx of every evaluation of its group through LDS, and rebuilds P (from lse,
 * with the dropout mask of (dropout_p, seed)) and dS (with delta, as the dq call wrote it) tile by tile:
 *   dv[dv_index[e]] (+)= P_drop^T dctx,   dk[dk_index[e]] (+)= dS^T Qs
 * k / v: the tile planes of the forward (kv_plane_stride = row pitch, kv_shape_stride in 16-bit elements), kv_index as in the
 * forward; a group's evaluations must share their key and their value slot (they do when grouped by output slot).  Run after
 * csn_block_attn_bwd_dq_recompute_f32 (probs_tiles = 0), which computes delta.  eval_ids / group_offsets / accumulate as in
 * the grouped csn_block_attn_bwd_dkv_f32 call; without group_offsets every listed evaluation is its own group. */
CSN_API int csn_block_attn_bwd_dkv_flash_f32(const float* dctx, long long ctx_eval_stride, const float* q, long long q_shape_stride,
                                     const int* q_index, const float* k, const float* v, long long kv_shape_stride,
                                     const int* kv_index, long long kv_plane_stride, int kv_f16, int ld, const float* lse,
                                     const float* delta, float* dk, float* dv, long long dkv_slot_stride,
                                     const int* dk_index, const int* dv_index, int accumulate, const int* eval_ids,
                                     int n_launch_evals, int n_heads, int d_head, int block, int n_blocks, int score_pitch,
                                     float dropout_p, unsigned long long seed, const int* group_offsets, int n_groups,
                                     void* stream);

/* ---- (3b) cross-length attention: one unchunked block per evaluation, n_queries != n_keys -------------
 * The MinkowskiNet variant of the layer (MinkowskiNet/models/attention.py:31-75, used per shape pair by
 * MinkowskiNet/models/hrnet.py:378-410, 456-470): softmax(Qs K^T) V over ALL keys of the other shape, gradients flowing
 * to queries, keys and values.  Same kernels as (2)/(3) with n_blocks = 1 and separate query / key counts:
 *   q, ctx, dctx, dq : [n_evals][n_heads*d_head][ld_q]   (n_queries <= ld_q)
 *   k, v, dk, dv     : [n_evals][n_heads*d_head][ld_kv]  (round-up-4(n_keys) <= ld_kv; the padding columns of dk / dv
 *                                                          are written as zeros)
 *   scores, dscores  : [n_evals][n_heads][n_queries][score_pitch], score_pitch >= round-up-4(n_keys) (32 in math mode 1)
 *   lse, delta       : [n_evals][n_heads][n_queries]
 * n_keys is arbitrary; n_queries must be a multiple of 4 (pad with zero points: their rows cost little and contribute
 * nothing to any gradient).  Evaluation e reads maps e (no slot indices, no accumulation). */
CSN_API int csn_cross_attn_fwd_f32(const float* q, const float* k, const float* v, long long q_shape_stride,
                           long long kv_shape_stride, int ld_q, int ld_kv, float* ctx, long long ctx_eval_stride,
                           float* scores, float* lse, int n_evals, int n_heads, int d_head, int n_queries, int n_keys,
                           int score_pitch, float rescale_threshold, float dropout_p, unsigned long long seed, void* stream);
CSN_API int csn_cross_attn_bwd_f32(const float* dctx, const float* ctx, long long ctx_eval_stride, const float* q, const float* k,
                           const float* v, long long q_shape_stride, long long kv_shape_stride, int ld_q, int ld_kv,
                           float* scores, float* dscores, const float* lse, float* delta, float* dq, float* dk, float* dv,
                           long long dq_eval_stride, long long dkv_eval_stride, int n_evals, int n_heads, int d_head,
                           int n_queries, int n_keys, int score_pitch, float dropout_p, unsigned long long seed,
                           void* stream);

/* ---- (3c) ragged batches of the cross-length attention ("varlen") -------------------------------------------
 * ONE launch chain for a batch of shape pairs whose query and key counts all differ — MinkowskiNet/models/hrnet.py:378-410,
 * 456-470 calls the layer once per shape / per shape pair, with 1..5 k voxels each.  The entry points of (3b) with length
 * arrays: n_queries[e] and n_keys[e] are DEVICE int arrays of n_evals entries (the cu_seqlens of a padded layout); max_queries
 * (% 4) and max_keys bound them and size the grid, score_pitch and the buffers, which are PADDED to common leading dimensions:
 *   q, ctx, dctx, dq [n_evals][n_heads*d_head][ld_q],  k, v, dk, dv [n_evals][n_heads*d_head][ld_kv],
 *   scores, dscores [n_evals][n_heads][max_queries][score_pitch],  lse, delta [n_evals][n_heads][max_queries].
 * n_queries[e] % 4 == 0 (round a shape's point count up: its zero points cost little and contribute nothing); n_keys[e] is
 * arbitrary (>= 1).  Work-groups of query tiles beyond n_queries[e] exit at once and key tiles beyond n_keys[e] are never
 * visited, so a short evaluation costs its own size.  Rows / columns beyond an evaluation's own counts are neither read as
 * data nor written: the caller zero-fills ctx, dq, dk, dv where it goes on to use the padding (csn_amd does), and the padding
 * points of the input maps must be finite (zeros).  Dropout masks are indexed with the launch's max_queries / score_pitch. */
CSN_API int csn_varlen_attn_fwd_f32(const float* q, const float* k, const float* v, long long q_shape_stride,
                            long long kv_shape_stride, int ld_q, int ld_kv, float* ctx, long long ctx_eval_stride,
                            float* scores, float* lse, int n_evals, int n_heads, int d_head, int max_queries, int max_keys,
                            const int* n_queries, const int* n_keys, int score_pitch, float rescale_threshold,
                            float dropout_p, unsigned long long seed, void* stream);
CSN_API int csn_varlen_attn_bwd_f32(const float* dctx, const float* ctx, long long ctx_eval_stride, const float* q, const float* k,
                            const float* v, long long q_shape_stride, long long kv_shape_stride, int ld_q, int ld_kv,
                            float* scores, float* dscores, const float* lse, float* delta, float* dq, float* dk, float* dv,
                            long long dq_eval_stride, long long dkv_eval_stride, int n_evals, int n_heads, int d_head,
                            int max_queries, int max_keys, const int* n_queries, const int* n_keys, int score_pitch,
                            float dropout_p, unsigned long long seed, void* stream);

/* ---- (4) output projection + residual + LayerNorm, forward -------------------------------------------
 * z[c][n] = sum_D wfc[c][D] ctx[e][D][n] + xres[res_index[e]][c][n];  xhat = (z - mean_c z) * rstd,
 * rstd = 1/sqrt(var_c z + eps).   Replaces fc + residual + LayerNorm (csa_models.py:52,57,114-118) up to the
 * LayerNorm's affine (gamma, beta), which the caller applies (it is needed un-applied by the backward).
 *   xhat [n_evals][d_model][ld],  rstd [n_evals][n_points].
 * dropout_p / seed: train-mode dropout on the fc output before the residual add (csa_models.py:56,115);
 * mask = function of (seed, evaluation, channel, point, ld) — the backward entry regenerates it from the same values.
 * 0 = eval mode.
 * xhat_sum (optional) [n_evals][d_model]: sum over the points of every xhat row — n_points * the pooled descriptor
 * mean_n SSA(x) of csa_models.py:211-212, 218-219 before the affine.  With a workspace sum_ws of
 * n_evals * ceil(n_points / 256) * d_model floats (sum_ws_floats says how many there are) the 256-channel bf16x3 kernel forms
 * per-tile partial sums in its epilogue and a small kernel adds them in a fixed order; without one, or on the other kernels,
 * a streaming pass over xhat follows (dense maps only: xhat_eval_stride == d_model * ld).
 * csn_outproj_ln_workspace_floats: the sum_ws size with which the fused sums are taken on every kernel that has them (the
 * streaming kernel of the bf16x3 mode keeps one partial per work-group and evaluation it touches; a smaller workspace is
 * never an error — the streaming pass runs instead). */
CSN_API long long csn_outproj_ln_workspace_floats(int n_evals, int d_model, int d_inner, int n_points);
CSN_API int csn_outproj_ln_fwd_f32(const float* ctx, long long ctx_eval_stride, const float* wfc, const float* xres,
                           long long xres_shape_stride, const int* res_index, float* xhat,
                           long long xhat_eval_stride, float* rstd, int n_evals, int d_model, int d_inner, int ld,
                           int n_points, float eps, float dropout_p, unsigned long long seed, float* xhat_sum,
                           float* sum_ws, long long sum_ws_floats, void* stream);

/* ---- (5) output projection + LayerNorm, backward -------------------------------------------------------
 * dz   = dropout-mask * LayerNorm-backward(dxhat; xhat, rstd)   [n_evals][d_model][ld]  (d fc output)
 * dz_res = the same without the mask (d residual input); may be NULL when input gradients are not needed
 * dctx = wfc^T dz                                          [n_evals][d_inner][ld]
 * dwfc (+)= sum_{e,n} dz[e][:,n] ctx[e][:,n]^T             [d_model][d_inner]
 * wfc_t is wfc transposed, [d_inner][d_model] row-major.  `ws` is scratch of at least
 * csn_wgrad_workspace_floats(d_model, d_inner, n_evals, n_points) floats.  accumulate != 0 adds into dwfc.
 * The incoming gradient is  dxhat[e][c][n] (evaluations e < n_dense_evals only; the others have none)
 *                         + dxhat_rows[e][c] (optional, NULL = none): a term that is constant along the points — the
 * gradient of the pooled means (csa_models.py:212,219) — so that it never has to be expanded to a full map.
 * dxhat_scale (optional) [n_dense_evals][d_model] and dxhat_group (>= 1; 0 = 1): the dense term of evaluation e is
 *   dxhat_scale[e][c] * dxhat[e / dxhat_group][c][n]
 * — with dxhat = the gradient of the mixed features (dfeats of csn_mix_bwd_f32, one map per query shape), dxhat_group = the
 * evaluations mixed per shape and dxhat_scale = comp * gamma, the per-evaluation gradient maps of the mix (:233, :238) are
 * rebuilt on the fly and never travel through HBM. */
CSN_API int csn_outproj_ln_bwd_f32(const float* dxhat, const float* xhat, const float* rstd, long long eval_stride,
                           const float* ctx, long long ctx_eval_stride, const float* wfc_t, float* dz, float* dz_res,
                           float* dctx, float* dwfc, float* ws, long long ws_floats, int n_evals, int d_model,
                           int d_inner, int ld, int n_points, int accumulate, float dropout_p,
                           unsigned long long seed, int dctx_split, long long dctx_plane_stride,
                           const float* dxhat_rows, int n_dense_evals, const float* dxhat_scale, int dxhat_group,
                           void* stream);

/* ---- (6) projection weight gradient ----------------------------------------------------------------------
 * dw[r][c] (+)= scale * sum_{s,n} dout[s][r][n] * x[s][c][n]        (autograd of csa_models.py:103-105) */
CSN_API int csn_project_wgrad_f32(const float* dout, long long dout_shape_stride, int ld_dout, const float* x,
                          long long x_shape_stride, int ld_x, float* dw, int rows, int channels, int n_shapes,
                          int n_points, float scale, int accumulate, float* ws, long long ws_floats,
                          void* stream);

/* Scratch floats needed by (5)/(6) for a [rows][cols] gradient reduced over n_maps maps of n_points points. */
CSN_API long long csn_wgrad_workspace_floats(int rows, int cols, int n_maps, int n_points);

/* ---- (7) retrieval measure for the shape kNN graph (csa_models.py:244-267) ------------------------------
 * r[i][j] = mean_n max_m cos(f1[i][n][:], f2[j][m][:]) over L2-normalised rows (eps 1e-12), for
 * POINT-MAJOR features f1 [s1][n1][channels], f2 [s2][n2][channels] as get_all_feats returns them
 * (csa_models.py:299).  ws: at least (s1*n1 + s2*n2) floats (inverse row norms) + s1*s2*n1 floats (per-point maxima): O(s1*s2),
 * so a caller with many shapes scores the query shapes in row chunks (csn_amd.functional.retrieval_measure does).  Any pair
 * count is accepted up to ceil(n1/128)*s1*s2 < 2^31 work-groups (CSN_E_ARG beyond). */
CSN_API int csn_retrieval_measure_f32(const float* f1, const float* f2, float* out, int s1, int n1, int s2, int n2,
                              int channels, float* ws, long long ws_floats, void* stream);

/* ---- (8) pooled descriptors and the cross-shape mix (csa_models.py:211-212, 218-219, 232-240) --------------
 * csn_rowsum_f32 : out[r] = sum_{n < n_points} x[r*ld + n]   (fp64 accumulation; the caller divides by n_points to
 *                  get the mean-over-points SSA descriptor of :212 / :219).
 * csn_mix_fwd_f32: feats[b][c][n] = gamma[c] * sum_k comp[b][k] * xhat[b*k1 + k][c][n] + beta[c] * sum_k comp[b][k]
 *                  i.e. sum_k comp_k * LayerNorm-affine(xhat_k): the compatibility-weighted sum of :233 and :238.
 * csn_mix_bwd_f32: dxhat[b*k1 + k][c][n] = comp[b][k] gamma[c] dfeats[b][c][n];
 *                  rowdot[b][k][c] = sum_n dfeats[b][c][n] xhat[b*k1+k][c][n];  rowsum[b][c] = sum_n dfeats[b][c][n]
 *                  (fp64 accumulation) from which d comp, d gamma, d beta follow with O(B*k1*C) host-side math.
 *                  dxhat (and dxhat_self) NULL: the reductions only — the maps are then rebuilt inside
 *                  csn_outproj_ln_bwd_f32 (dxhat_scale / dxhat_group).
 * xhat_self / dxhat_self != NULL: the k = 0 maps (the shape's own evaluation) live in their own tensors [b][c][n] and
 * xhat / dxhat hold the k1 - 1 others, [b*(k1-1) + k-1] — the form the overlapped multi-GPU path produces (own shapes are
 * evaluated while the neighbour exchange is in flight), so that no concatenation of the two is ever built.
 * All maps dense channel-major [..][channels][n_points], n_points % 4 == 0, k1 <= 8. */
CSN_API int csn_rowsum_f32(const float* x, float* out, long long rows, int n_points, long long ld, void* stream);
CSN_API int csn_mix_fwd_f32(const float* xhat, const float* comp, const float* gamma, const float* beta, float* feats,
                    int n_shapes, int k1, int channels, int n_points, const float* xhat_self, void* stream);
CSN_API int csn_mix_bwd_f32(const float* dfeats, const float* xhat, const float* comp, const float* gamma, float* dxhat,
                    float* rowdot, float* rowsum, int n_shapes, int k1, int channels, int n_points, const float* xhat_self,
                    float* dxhat_self, void* stream);

/* ---- (9) the compatibility head (csa_models.py:222-230) -----------------------------------------------------
 * comp[b][k] = softmax_k < normalize(wq y[b][0] + bq), normalize(wk key(b, k) + bk) >  over the k1 = K+1 pooled descriptors
 * y (n_shapes, k1, channels) of every query shape (the shape itself in slot 0), F.normalize's eps = 1e-12, channels <= 256,
 * k1 <= 8.  reference_layout != 0: the key rows are taken the way the reference's bookkeeping takes them for B > 1 — the key
 * descriptors stacked neighbour-major and re-viewed as (B, k1, C) (csa_models.py:213,220,227): key(b, k) = y[(b k1 + k) % B]
 * [(b k1 + k) / B]; 0: key(b, k) = y[b][k].
 * Every sum accumulates in fp64 (the head's gradients are differences of nearly equal descriptors; 63 MFLOP).
 * Forward: wq_t / wk_t are the nn.Linear weights TRANSPOSED ([in][out]); save_u (n_shapes, k1 + 1, channels) and save_norm
 * (n_shapes, k1 + 1) keep the normalised vectors and the norms for the backward, as DOUBLES.
 * Backward: wq / wk as stored ([out][in]); ws >= 2 n_shapes (k1 + 1) channels DOUBLES of scratch; writes the gradients
 * dpooled (n_shapes, k1, channels), dwq / dwk (channels, channels) and dbq / dbk (channels) — overwritten, not accumulated;
 * sums over the shapes run in a fixed order (bitwise reproducible). */
CSN_API int csn_compat_fwd_f32(const float* pooled, const float* wq_t, const float* bq, const float* wk_t, const float* bk, float* comp,
                       double* save_u, double* save_norm, int n_shapes, int k1, int channels, int reference_layout, void* stream);
CSN_API int csn_compat_bwd_f32(const float* dcomp, const float* comp, const double* save_u, const double* save_norm, const float* pooled,
                       const float* wq, const float* wk, double* ws, long long ws_doubles, float* dpooled, float* dwq, float* dbq,
                       float* dwk, float* dbk, int n_shapes, int k1, int channels, int reference_layout, void* stream);

/* ---- (10) the loss the layers are trained with (csa_training.py:94-108) -----------------------------------------------
 * The reference transposes the logits to [point][class], gathers the points with label > mask and calls F.cross_entropy and
 * an argmax accuracy on the selection.  Here the class-major logits [shape][class][ld] (the layout the logit layer writes,
 * csa_models.py:151) are read where they lie:
 *   forward   stats[0] = mean over the counted points (mask < label < n_classes) of lse - z[label], stats[1] = the fraction of
 *             them whose first arg-max class is the label, stats[2] = their number (0 counted points: 0 / 0 = nan, as the
 *             reference's empty selection gives); lse [n_shapes][n_points] is kept for the backward.  ws: scratch of
 *             csn_masked_ce_workspace_bytes(n_shapes, n_points) bytes, 8-byte aligned (per-block fp64 partial sums, added in a
 *             fixed order: bitwise reproducible).
 *   backward  dlogits[s][c][n] = counted ? (exp(z - lse) - [c == label]) * grad_out[0] / stats[2] : 0, every class row of
 *             every shape written; any point count, pitch and alignment (16-byte accesses where n_points, ld, dld and the shape
 *             strides are % 4 == 0 and logits, lse, dlogits 16-byte aligned; one point per thread otherwise).  A logit of -inf
 *             is a probability of zero (also as a point's first class).
 * labels are int64 (torch.long), label_shape_stride elements apart per shape. */
CSN_API long long csn_masked_ce_workspace_bytes(int n_shapes, int n_points);
CSN_API int csn_masked_ce_fwd_f32(const float* logits, long long shape_stride, int ld, const long long* labels, long long label_shape_stride,
                          int n_shapes, int n_classes, int n_points, int mask, float* lse, void* ws, long long ws_bytes,
                          float* stats, void* stream);
CSN_API int csn_masked_ce_bwd_f32(const float* logits, long long shape_stride, int ld, const long long* labels, long long label_shape_stride,
                          int n_shapes, int n_classes, int n_points, int mask, const float* lse, const float* stats,
                          const float* grad_out, float* dlogits, long long dshape_stride, int dld, void* stream);

/* ---- DEVELOPMENT SECTION -------------------------------------------------------------------------------------------------
 * Kernel-selection switches for A/B timing and for the equality tests between two kernel forms of one product.  They are
 * PROCESS-wide, not thread-safe, change no result beyond fp32 rounding and are not part of the drop-in surface: a product
 * build leaves every key at its default.  csn_dev_set returns the previous value (CSN_E_ARG for an unknown key).
 *   CSN_DEV_BIG_TILES   1   256 x 256 GEMM tiles where the output fills them (0: 128 x 128 tiles everywhere)
 *   CSN_DEV_WIDE_GEMM   1   sixteen-wave form of the 256 x 256 tiles in the bf16x3 mode (0 off, 2: the one-plane modes too)
 *   CSN_DEV_WIDE_FORMS  7   bit set of the product forms that take it: 1 plain, 2 tile-plane B (dV / dK), 4 weight gradients
 *   CSN_DEV_WX          9   bit set for the K = 256 weight products of the bf16x3 mode: 1 projections, dCtx and out-projection +
 *                           LayerNorm on the weight-stationary streaming kernel (0: the tiled GEMM kernels); 2 its wave halves
 *                           staggered; 4 out-projection + LayerNorm back on the tiled kernel; 8 LayerNorm backward fused into
 *                           the dCtx stream; bits 4..7: timing-only ablations, results wrong */
#define CSN_DEV_BIG_TILES 0
#define CSN_DEV_WIDE_GEMM 1
#define CSN_DEV_WIDE_FORMS 2
#define CSN_DEV_WX 3
/* (key 4 was the 32-queries-per-wave attention forward of round 4: measured 25 % slower, removed; profiles/README.md) */
#define CSN_DEV_LNB_GROUP 5 /* default 0; G > 0: csn_outproj_ln_bwd_f32 runs its LayerNorm backward and its dCtx product over groups
                               of G evaluations (bf16x3, streaming dCtx; the same results) */
/* (key 6 was the output-stationary dV / dK stream of round 5: the GEMM route is as fast over the step; profiles/r5_dkv_stream.txt) */
CSN_API int csn_dev_set(int key, int value);
CSN_API int csn_dev_get(int key);

#ifdef __cplusplus
}
#endif
#endif /* CSN_HIP_H */
