// Internal launch interface between the kernel translation units and the C-ABI (csn_capi.hip).
#pragma once
#include "csn_common.h"

struct CsnGemmArgs {
  CsnOperand A, B, C;
  int M, N, K;
  int n0, n1;        // blockIdx.z = (z2 * n1 + z1) * n0 + z0
  int k_chunk;       // > 0: this z0 contracts only k in [0, min(k_chunk, K - z0*k_chunk)); the operand
                     //      strides s0 carry the matching offsets (split-K over the point index)
  float alpha;
  int div_rows;      // output rows m < div_rows are divided by div_val (query scaling, csa_models.py:139)
  float div_val;
  int accumulate;    // C += result
  const int* eval_ids;   // optional: blockIdx.z's slowest index z2 -> evaluation id, applied before the operands' idx2
  int batch;             // number of batch items z (filled in by the launcher)
  // grouped accumulation (256 x 256 bf16x3 kernel): the slowest batch index z2 counts GROUPS; group g contracts the items
  // grp_items[grp_off[g] .. grp_off[g+1]) one after the other into the same accumulators (A and B are addressed by the item,
  // C by the group's first item) and writes once — evaluations that share an output slot need no read-modify-write passes
  const int* grp_off = nullptr;
  const int* grp_items = nullptr;
  // ragged batches (varlen attention backward): per-item column / contraction counts, indexed by the slowest batch index
  // z2 (after eval_ids); N and K above are then the maxima that size the grid.  n_arr[z] is rounded up to 4; k_arr[z] % 4 == 0.
  const int* n_arr = nullptr;
  const int* k_arr = nullptr;
  // ragged last block (block attention backward): the LAST item of the fastest batch index z0 (= the last attention block)
  // has n_last columns and k_last contraction steps (0 = like the others)
  int n_last = 0, k_last = 0;
};

int csn_launch_gemm_f32(const CsnGemmArgs& a, int b_is_nk, int batch, hipStream_t st);
int csn_launch_gemm_bf16x3(const CsnGemmArgs& a, int b_is_nk, int batch, int mode, hipStream_t st);   // gemm_bf16x3.hip; mode 1 bf16x3, 2 bf16, 3 fp16
int csn_gemm_tile_major_planes(int M, int N);         // 1: B.planes == 3 (tile-major tile planes, CsnAttnArgs::sc_layout) is available for an M x N output
int csn_gemm_bf16x3_big_tiles(int M, int N);          // 1: an M x N output takes the 256 x 256 kernel (grouped accumulation available)
int csn_launch_slab_reduce(const float* slab, float* out, int n_slabs, long long n, float alpha, int accumulate,
                           hipStream_t st);

// ---- weight-stationary streaming products, K = 256, bf16x3 (wx_stream.hip) ---------------------------------
struct CsnWxArgs {
  const float* w;                                         // [n_sets * 256][256] row-major
  const float* x;  long long x_item_stride;  int ldx;     // [item][256][ldx] fp32
  void* out;  long long out_item_stride;  int ldo;        // fp32 [item][n_sets * 256][ldo], or bf16 tile planes (strides in 16-bit elements)
  int n_items, n_points, n_sets;
  int div_rows;  float div_val;  float div_rcp;  int div_exact;   // rows < div_rows are divided by div_val (exact as * div_rcp when it is a power of two)
  int tb;                                                 // tile planes: points per attention block
  int stagger = 0;                                        // waves 4..7 half an iteration behind waves 0..3 (development; filled in by the launcher)
  int ablate = 0;                                         // development: timing-only ablations (bits 4..7 of CSN_DEV_WX)
  // out_mode 3 — out-projection + fc dropout + residual + LayerNorm (csa_models.py:115-118): x = Ctx^T, w = W_fc, out = xhat
  const float* res = nullptr;  long long res_shape_stride = 0;  const int* res_index = nullptr;   // residual x[shape][256][ldo]
  float* rstd = nullptr;                                  // [item][n_points]
  float eps = 0.f, dropout_p = 0.f;  unsigned long long seed = 0;
  float* sum_ws = nullptr;  int sum_slots = 0;            // optional [item][sum_slots][256]: per-stream sums over points of xhat
  // out_mode 4 — Q | K | V in one pass over x: the first n_f32 row sets leave as fp32 maps, the others as tile planes through `out`
  float* out_f32 = nullptr;  long long out_f32_item_stride = 0;  int ldo_f32 = 0;  int n_f32 = 0;
};
// "this geometry is not taken by the streaming kernel": an INTERNAL return value of the csn_launch_wx* launchers, distinct from
// every CSN_E_* status of the C ABI — the caller falls through to the tiled kernels and never hands it to the user
constexpr int CSN_NOT_TAKEN = -1000;
extern int csn_gemm_big_tiles, csn_gemm_wide, csn_gemm_wide_set, csn_dev_wx, csn_dev_lnb_group;   // development switches (csn_dev_set)
bool csn_wx_geometry_takes(int n_items, int n_points, int n_sets);   // ... and this launch geometry (else CSN_NOT_TAKEN)
bool csn_wx_takes(int rows, int k);                       // this product shape runs on the streaming kernel
int csn_launch_wx(const CsnWxArgs& a, int out_mode /* 0 fp32, 2 tile planes, 3 LayerNorm, 4 fp32 + tile planes */, hipStream_t st);
int csn_wx_ln_sum_slots(int n_items, int n_points);      // out_mode 3: sum_slots the launch will use (sum_ws = n_items * slots * 256 floats)
int csn_launch_wx_ln_sums(const float* ws, float* out, int n_items, int n_points, hipStream_t st);   // out[item][256] from sum_ws

// ---- fused block attention (attn_f32.hip) -----------------------------------------------------
struct CsnAttnArgs {
  // channel-major projected features, one slab of [H*d][ld] per shape
  const float* q;  const float* k;  const float* v;     // forward: Qs^T, K^T, V^T ; backward: dO^T(=dCtx^T), V^T, K^T
  long long q_shape_stride, kv_shape_stride;             // elements between consecutive shapes
  const int* q_index; const int* kv_index;               // evaluation e -> shape slot (nullptr: identity)
  int ld;                                                // points per row (leading dimension)
  float* out;        long long out_eval_stride;          // forward: Ctx^T[e][H*d][ld] ; backward: dQs^T[e][H*d][ld]
  float* scores;                                         // S^T / P^T  [e][h][blk][T][Tp]   (may be null in forward)
  float* dscores;                                        // backward: dS^T, same geometry
  float* lse;                                            // [e][h][n_blocks*T]
  float* delta;                                          // backward: rowsum(dO*O) [e][h][n_blocks*T], computed by the kernel
  const float* ctx;                                      // backward: O^T = Ctx^T [e][H*d][ld] of the forward (for delta)
  int E, H, T, Tp, n_blocks;
  float rescale_threshold;
  const int* eval_ids;                                   // launch z -> evaluation id (nullptr: identity); E = launch size
  const int* grp_off;                                    // backward, optional: E counts groups; group g = eval_ids[grp_off[g] .. grp_off[g+1])
  const int* out_index;                                  // evaluation -> output slot (nullptr: the evaluation itself)
  int accumulate;                                        // out += result (several evaluations share an output slot)
  float dropout_p;                                       // attention-probability dropout (csa_models.py:141); 0 = off
  unsigned long long seed;
  // bf16x3 kernels only.  kv_planes != 0: k and v point into a "tile plane" tensor written by the projection
  // (bf16; per row and block 16 tiles of [hi: 32 keys | lo: 32 keys], block pitch 1024, row pitch kv_ld = n_blocks * 1024);
  // kv_shape_stride then counts bf16 elements.  r_planes / *_plane_stride: reserved.
  int r_planes, kv_planes;
  long long r_plane_stride, kv_plane_stride;
  int kv_ld;
  int Tq;                                                // queries per block (0: = T); T counts the keys
  int ld_kv;                                             // leading dimension of the fp32 K/V maps (0: = ld)
  int sc_tiles;                                          // backward (bf16x3): P and dS leave as bf16 tile planes [query][tile: hi 32 | lo 32]
  // ragged batches (n_blocks = 1): per-evaluation query / key counts (device arrays; null = Tq / T for every evaluation).
  // Tq and T are then the maxima: they size the grid, the score pitch and the buffers; tq_arr[e] % 4 == 0.
  const int* tq_arr;
  const int* t_arr;
  int T_last;                                            // block mode: queries = keys of the LAST block (0: = T) — a row that ends inside it
  // backward with score recomputation (16-bit modes, tile-plane K / V): the pre-scaled queries Qs^T [slot][H*d][ld] of the
  // forward; q2 == nullptr = the scores are read from `scores` (kept by the forward)
  const float* q2 = nullptr; long long q2_shape_stride = 0; const int* q2_index = nullptr;
  int kv_f16 = 0;                                        // backward, one-plane mode: the K / V tile planes hold fp16 (the forward ran in math mode 3)
  // 16-bit activation maps (single-product modes): format of the register operand q (forward: Qs; backward: dO), of ctx (O),
  // of q2 and of out — 0 fp32, 1 bf16, 2 fp16; shape / evaluation strides then count 16-bit elements
  int r_fmt = 0, ctx_fmt = 0, q2_fmt = 0, out_fmt = 0;
  // score storage (block mode, two planes, tile-plane K / V): 0 = [query][key] rows of pitch Tp; 1 = TILE-MAJOR, per block
  // [key tile kt][query][32 keys] (the 128 bytes of a query's tile, fp32 scores or [hi 32 | lo 32] planes): what a wave stores
  // or loads per instruction is then one contiguous run instead of sixteen 64-byte pieces 2 KB apart
  int sc_layout = 0;
};
// score recomputation needs three LDS tile images per stage: one plane at every width, two planes up to d = 128
// (-DCSN_RC_ALIAS=1, TIMING EXPERIMENT ONLY — results are wrong: the two-plane d = 256 instance is built with its third image
//  laid over the second, to price a one-K-image recomputing kernel before writing it; profiles/r4r_recompute_d256_experiment.txt)
#ifndef CSN_RC_ALIAS
#define CSN_RC_ALIAS 0
#endif
constexpr bool csn_attn_recompute_fits(int planes, int dt) { return planes == 1 || dt <= 4 || CSN_RC_ALIAS; }
// ---- key-stationary dK / dV with recomputed scores (attn_dkv.hip; 16-bit modes, block mode, d <= 128) ----------------
struct CsnAttnDkvArgs {
  const float* q;     long long q_shape_stride;  const int* q_index;     // pre-scaled queries Qs^T [slot][H*d][ld], evaluation -> slot
  const float* dctx;  long long ctx_eval_stride;                          // dO^T [evaluation][H*d][ld]
  const float* k;  const float* v;                                        // tile planes (16-bit) of the slots' keys / values
  long long kv_shape_stride;  int kv_ld;  const int* kv_index;            // in 16-bit elements; evaluation -> key/value slot
  const float* lse;  const float* delta;                                  // [evaluation][H][n_blocks * T]
  float* dk;  float* dv;  long long dkv_slot_stride;                      // fp32 gradient maps [slot][..][ld]
  const int* dk_index;  const int* dv_index;                              // evaluation -> output slot (nullptr: the evaluation)
  int accumulate;
  const int* eval_ids;  const int* grp_off;  int n_groups;                // group g = eval_ids[grp_off[g] .. grp_off[g+1]); no offsets: one evaluation each
  int ld, H, T, Tp, n_blocks, T_last;
  float dropout_p;  unsigned long long seed;
  int kv_f16 = 0;                                                         // one-plane mode: k / v hold fp16 (forward of math mode 3)
  int q_fmt = 0, dctx_fmt = 0;                                            // 16-bit activation maps: 0 fp32, 1 bf16, 2 fp16 (q of a mode-3 forward)
  int out_fmt = 0;                                                        // 1: dk / dv leave as bf16 maps (written once: no accumulate)
};
int csn_launch_attn_dkv_flash(const CsnAttnDkvArgs& a, int d, int mode, hipStream_t st);
constexpr bool csn_attn_dkv_flash_fits(int dt) { return dt <= 4; }         // K^T, V^T, dK^T, dV^T of 16 keys in one wave's registers

int csn_launch_attn_fwd_f32(const CsnAttnArgs& a, int d, hipStream_t st);
int csn_launch_attn_bwd_f32(const CsnAttnArgs& a, int d, hipStream_t st);
int csn_launch_attn_fwd_bf16x3(const CsnAttnArgs& a, int d, int mode, hipStream_t st);     // attn_bf16x3.hip; mode 1..3 (2, 3: tile-plane K/V only)
int csn_launch_attn_bwd_bf16x3(const CsnAttnArgs& a, int d, int mode, hipStream_t st);     // mode 1, 2

// ---- output projection + residual + LayerNorm (outproj_ln.hip) --------------------------------
struct CsnOutProjArgs {
  const float* ctx;   long long ctx_eval_stride;         // Ctx^T[e][D][ld]
  const float* wfc;                                      // [C][D] row-major (csa_models.py:52)
  const float* xres;  long long xres_shape_stride; const int* res_index;   // residual x[shape][C][ld] (:99,:116)
  float* xhat;        long long xhat_eval_stride;        // normalised output [e][C][ld]
  float* rstd;                                           // [e][n_points]
  int E, C, D, ld, n_points;
  float eps;
  float dropout_p;                                       // dropout on the fc output (csa_models.py:115); 0 = off
  unsigned long long seed;
  // optional: xhat_sum[e][c] = sum over points of xhat (the pooled descriptor of csa_models.py:211-212 before the affine).
  // The 256 x 256-tile kernel forms per-tile partial sums in its epilogue (sum_ws [e][ceil(n_points/256)][C], reduced by a
  // small second kernel in fp64); the other kernels are followed by the streaming row-sum pass.
  float* xhat_sum; float* sum_ws; long long sum_ws_floats;
  // 16-bit activation maps (math modes 2 / 3): ctx is a 16-bit map of the mode's type, xhat leaves as fp16; strides in elements
  int act16 = 0;
};
int csn_launch_partial_sums_f32(const float* ws, float* out, long long rows_outer, int tiles, int C, hipStream_t st);
int csn_launch_outproj_ln_fwd_f32(const CsnOutProjArgs& a, int mode, hipStream_t st);   // mode 0: fp32, 1..3: 16-bit matrix-core contraction
int csn_launch_outproj_ln_big(const CsnOutProjArgs& a, int mode, hipStream_t st);       // gemm_bf16x3.hip: C = 256, 256 x 256 tiles

// masked cross-entropy on class-major logits (loss.hip; csa_training.py:94-108)
struct CsnMaskedCeArgs {
  const float* logits;  long long shape_stride;  int ld;          // [shape][class][ld]
  const long long* labels;  long long label_shape_stride;         // [shape][n_points] int64
  int n_shapes, n_classes, n_points, mask;                        // counted: mask < label < n_classes
  float* lse;                                                     // [shape][n_points]
  double* partials;                                               // forward: 3 per block (csn_masked_ce_blocks)
  float* stats;                                                   // [3]: mean loss, accuracy, counted points
  const float* grad_out;  float* dlogits;  long long dshape_stride;  int dld;      // backward
};
long long csn_masked_ce_blocks(int n_shapes, int n_points);
int csn_launch_masked_ce_fwd(const CsnMaskedCeArgs& a, hipStream_t st);
int csn_launch_masked_ce_bwd(const CsnMaskedCeArgs& a, hipStream_t st);

struct CsnLnBwdArgs {
  const float* dxhat; const float* xhat; const float* rstd;   // [e][C][ld], [e][C][ld], [e][n_points]
  const float* dxhat_rows;                                     // optional [e][C]: added to every point of row (e, c)
  int n_dense;                                                 // evaluations e >= n_dense have no dense dxhat (only the row term)
  // dense part of evaluation e < n_dense = dxhat_scale[e][c] * dxhat[e / dxhat_group][c][n]  (scale null: 1; group 1: map e).
  // With group = K+1 and scale = comp * gamma the maps are the gradient of the mixed features themselves: the per-evaluation
  // gradient maps of the mix are never written
  const float* dxhat_scale; int dxhat_group;
  float* dz;                                                   // [e][C][ld]  gradient w.r.t. the fc output (dropout mask applied)
  float* dz_res;                                               // optional: gradient w.r.t. the residual input (no mask)
  long long eval_stride;
  int E, C, ld, n_points;
  float dropout_p;
  unsigned long long seed;
  int act16 = 0;                                               // xhat is an fp16 map, dz leaves as a bf16 map (dxhat, dz_res stay fp32)
  int e_base = 0;                                              // the launch covers evaluations e_base .. e_base + E - 1 of the maps
};
int csn_launch_ln_bwd_f32(const CsnLnBwdArgs& a, hipStream_t st);
// LayerNorm backward fused into the dCtx stream (wx_lnb.hip; bf16x3, d_model = d_inner = 256, fp32 maps)
struct CsnWxLnbArgs {
  const float* w;                                              // W_fc^T [d_inner][d_model] row-major
  const float* xhat; const float* rstd; long long eval_stride; int ld;
  const float* dxhat; int dxhat_group; int n_dense; const float* dxhat_scale; const float* dxhat_rows;     // as CsnLnBwdArgs
  float* dz; float* dz_res;                                    // [e][256][ld]; dz_res optional
  float* dctx; long long dctx_eval_stride;                     // [e][256][ld]
  int n_items, n_points, e_base;                               // evaluations e_base .. e_base + n_items - 1
  float dropout_p; unsigned long long seed;
  int ablate = 0;
};
bool csn_wx_lnb_takes(const CsnLnBwdArgs& a, int d_inner);
int csn_launch_wx_lnb(const CsnWxLnbArgs& a, hipStream_t st);

// delta[e][h][n] = sum_{c in head h} a[e][c][n] * b[e][c][n]
int csn_launch_rowdot_f32(const float* a, const float* b, float* out, const int* eval_ids, int E, int H, int d, int ld,
                          int n_points, long long eval_stride, int a_split, long long a_plane_stride, hipStream_t st);

// ---- pooled descriptors / cross-shape mix (combine.hip) ---------------------------------------------
// x16 != 0: the normalised maps (x / xhat / xhat0) are fp16 (16-bit activation maps)
int csn_launch_rowsum_f32(const float* x, float* out, long long rows, int n, long long ld, hipStream_t st, int x16 = 0);
int csn_launch_mix_fwd_f32(const float* xhat, const float* comp, const float* gamma, const float* beta, float* feats, int B,
                           int K1, int C, int NP, const float* xhat0, hipStream_t st, int x16 = 0);
int csn_launch_mix_bwd_f32(const float* dfeats, const float* xhat, const float* comp, const float* gamma, float* dxhat,
                           float* rowdot, float* rowsum, int B, int K1, int C, int NP, const float* xhat0, float* dxhat0,
                           hipStream_t st, int x16 = 0);
int csn_launch_retrieval_f32(const float* f1, const float* f2, float* out, int s1, int n1, int s2, int n2, int C,
                             float* ws, hipStream_t st);

// ---- compatibility head (compat.hip): normalize(W_q y_0 + b), normalize(W_k y_k + b), dot, softmax over the K+1 keys ----
int csn_launch_compat_fwd(const float* pooled, const float* wq_t, const float* bq, const float* wk_t, const float* bk, float* comp,
                          double* save_u, double* save_n, int B, int K1, int C, int reference_layout, hipStream_t st);
int csn_launch_compat_bwd(const float* dcomp, const float* comp, const double* save_u, const double* save_n, const float* pooled,
                          const float* wq, const float* wk, double* d_raw, double* dx, float* dpooled, float* dwq, float* dbq,
                          float* dwk, float* dbk, int B, int K1, int C, int reference_layout, hipStream_t st);
