// extern "C" surface of libcsn_hip.so (declared in include/csn_hip.h).  Argument checking and the
// decomposition of each domain-level call into kernel launches live here; no allocation, no sync.
#include "../../include/csn_hip.h"
#include "csn_kernels.h"
#include <cmath>

namespace {

// 0: exact fp32 matrix cores; 1: bf16x3 (three bf16 products per fp32 product); 2: bf16; 3: fp16 (one product)
int g_math_mode = CSN_MATH_BF16X3;          // process default (csn_set_math_mode)
thread_local int t_math_mode = -1;          // per-thread override (csn_set_thread_math_mode); -1 = none

inline int mode() { return t_math_mode >= 0 ? t_math_mode : g_math_mode; }
// 16-bit activation maps between the entry points (csn_set_thread_act16): 0 off, 1 = the forward's maps are bf16 (math mode 2),
// 2 = they are fp16 (math mode 3 forward; its backward runs in mode 2 and converts them while staging)
// + 4: the gradient maps dQ / dK / dV (and what reads them: the projection weight gradients, dx) are bf16 maps too
thread_local int t_act16 = 0;
inline int act16() { return t_act16 & 3; }
inline bool grad16() { return (t_act16 & 4) != 0; }
// forward entry points: the flag has to name the type of the mode that runs; backward: mode 2 takes either
inline bool act16_fwd_ok() { return act16() == 0 || (act16() == 1 && mode() == 2) || (act16() == 2 && mode() == 3); }
inline bool act16_bwd_ok() { return act16() == 0 || mode() == 2; }
// score storage of the block-attention entry points (csn_set_thread_score_layout): 0 = [query][key] rows, 1 = tile-major
thread_local int t_score_layout = 0;
inline int planes_of(int m) { return m == 1 ? 2 : 1; }          // tile-plane / split-tensor planes of a 16-bit mode

// the cross-length entry points (fp32 K / V maps) have no single-product kernels: in modes 2 / 3 they run as mode 1
struct ModeGuard {
  int saved;
  explicit ModeGuard(int m) : saved(t_math_mode) { t_math_mode = m; }
  ~ModeGuard() { t_math_mode = saved; }
};

int launch_gemm(const CsnGemmArgs& a, int b_is_nk, int batch, hipStream_t st) {
  if (mode() != 0) {
    if (a.M <= 0 || a.N <= 0 || batch <= 0) return 0;
    if ((a.A.ld & 3) || (a.B.ld & 3) || (a.K & 3) || (a.k_chunk & 3)) return CSN_E_ALIGN;
    if (!b_is_nk && (a.N & 3)) return CSN_E_ALIGN;
    return csn_launch_gemm_bf16x3(a, b_is_nk, batch, mode(), st);
  }
  return csn_launch_gemm_f32(a, b_is_nk, batch, st);
}

inline bool mis16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

// K = 256 weight products of the bf16x3 mode on the weight-stationary streaming kernel (wx_stream.hip)
int launch_wx(const float* w, const float* x, long long x_stride, int ldx, void* out, long long out_stride, int ldo, int rows,
              int n_items, int n_points, int div_rows, float div_val, int out_mode, int tb, hipStream_t st) {
  CsnWxArgs a;
  a.w = w; a.x = x; a.x_item_stride = x_stride; a.ldx = ldx;
  a.out = out; a.out_item_stride = out_stride; a.ldo = ldo;
  a.n_items = n_items; a.n_points = n_points; a.n_sets = rows / 256;
  a.div_rows = div_rows; a.div_val = div_val; a.div_rcp = 1.f / div_val;
  int ex = 0;
  a.div_exact = (std::frexp(div_val, &ex) == 0.5f && div_val > 0.f) ? 1 : 0;     // a power of two: v * (1 / t) == v / t bit for bit
  a.tb = tb;
  return csn_launch_wx(a, out_mode, st);
}

// Block mode, n_blocks * block > ld: the row of ld points ends inside the last block, which then holds
// ld - (n_blocks - 1) * block points (a multiple of 4).  Returns that count, 0 for full blocks, or a negative status.
inline int last_block_points(int block, int n_blocks, int ld, int block_q) {
  if (block_q != 0) return 0;                                  // cross-length entries: key and query counts are explicit
  if ((long long)n_blocks * block <= ld) return 0;
  if ((long long)(n_blocks - 1) * block >= ld) return CSN_E_ARG;
  const int t = ld - (n_blocks - 1) * block;
  return (t & 3) ? CSN_E_ALIGN : t;
}
inline bool dim_ok(int d) { return d == 32 || d == 64 || d == 96 || d == 128 || d == 256; }

CsnOperand operand(const float* p, long long s0, long long s1, long long s2, const int* idx2, int ld) {
  CsnOperand o;
  o.ptr = const_cast<float*>(p);
  o.s0 = s0; o.s1 = s1; o.s2 = s2; o.idx2 = idx2; o.ld = ld;
  o.planes = 0; o.plane_stride = 0;
  return o;
}

// points contracted per work-group of a split weight gradient
int wgrad_chunk(int n_maps, int n_points) {
  // ~512 equal slabs: every map is cut into the same number of equal chunks (unequal tails leave CUs idle at the end)
  long long total = (long long)n_maps * n_points;
  long long want = (total + 511) / 512;
  if (want < 256) want = 256;
  long long n_chunks = (n_points + want - 1) / want;
  if (n_chunks < 1) n_chunks = 1;
  long long c = (n_points + n_chunks - 1) / n_chunks;
  c = ((c + 3) / 4) * 4;
  return (int)c;
}

// dw[rows][cols] (+)= scale * sum_{z2, n} a[z2][rows][n] * b[z2][cols][n]      (a_fmt / b_fmt: CSN_FMT_* of the two maps)
int wgrad(const float* a, long long a_stride, int lda, const float* b, long long b_stride, int ldb, float* dw, int rows,
          int cols, int n_maps, int n_points, float scale, int accumulate, float* ws, long long ws_floats,
          hipStream_t st, int a_fmt = 0, int b_fmt = 0) {
  const int chunk = wgrad_chunk(n_maps, n_points);
  const int n_chunks = (n_points + chunk - 1) / chunk;
  const long long slabs = (long long)n_maps * n_chunks;
  if (ws_floats < slabs * rows * cols) return CSN_E_WORKSPACE;
  CsnGemmArgs g;
  g.A = operand(a, chunk, 0, a_stride, nullptr, lda);
  g.B = operand(b, chunk, 0, b_stride, nullptr, ldb);
  g.A.fmt = a_fmt; g.B.fmt = b_fmt;
  g.C = operand(ws, (long long)rows * cols, 0, (long long)n_chunks * rows * cols, nullptr, cols);
  g.M = rows; g.N = cols; g.K = n_points;
  g.n0 = n_chunks; g.n1 = 1; g.k_chunk = chunk;
  g.alpha = 1.f; g.div_rows = 0; g.div_val = 1.f; g.accumulate = 0; g.eval_ids = nullptr;
  int rc = launch_gemm(g, /*b_is_nk=*/1, (int)slabs, st);
  if (rc) return rc;
  return csn_launch_slab_reduce(ws, dw, (int)slabs, (long long)rows * cols, scale, accumulate, st);
}

}  // namespace

extern "C" {

int csn_version(void) { return CSN_ABI_VERSION; }

int csn_dev_get(int key) {
  switch (key) {
    case CSN_DEV_BIG_TILES: return csn_gemm_big_tiles;
    case CSN_DEV_WIDE_GEMM: return csn_gemm_wide;
    case CSN_DEV_WIDE_FORMS: return csn_gemm_wide_set;
    case CSN_DEV_WX: return csn_dev_wx;
    case CSN_DEV_LNB_GROUP: return csn_dev_lnb_group;
    default: return CSN_E_ARG;
  }
}
int csn_dev_set(int key, int value) {
  const int prev = csn_dev_get(key);
  switch (key) {
    case CSN_DEV_BIG_TILES: csn_gemm_big_tiles = value; break;
    case CSN_DEV_WIDE_GEMM: csn_gemm_wide = value; break;
    case CSN_DEV_WIDE_FORMS: csn_gemm_wide_set = value; break;
    case CSN_DEV_WX: csn_dev_wx = value; break;
    case CSN_DEV_LNB_GROUP: csn_dev_lnb_group = value < 0 ? 0 : value; break;
    default: return CSN_E_ARG;
  }
  return prev;
}

int csn_set_math_mode(int m) {
  if (m < 0 || m > 3) return CSN_E_ARG;
  g_math_mode = m;
  return 0;
}
int csn_set_thread_math_mode(int m) {
  if (m < -1 || m > 3) return CSN_E_ARG;
  t_math_mode = m;
  return 0;
}
int csn_get_math_mode(void) { return mode(); }
int csn_get_thread_math_mode(void) { return t_math_mode; }
int csn_set_thread_act16(int fmt) {
  if (fmt < 0 || (fmt & 3) == 3 || fmt > 6 || fmt == 4) return CSN_E_ARG;
  t_act16 = fmt;
  return 0;
}
int csn_get_thread_act16(void) { return t_act16; }
int csn_set_thread_score_layout(int layout) {
  if (layout < 0 || layout > 1) return CSN_E_ARG;
  t_score_layout = layout;
  return 0;
}
int csn_get_thread_score_layout(void) { return t_score_layout; }

const char* csn_status_string(int status) {
  switch (status) {
    case 0: return "ok";
    case CSN_E_ARG: return "csn: null pointer, non-positive size, a count beyond its row, or a flag this math mode does not take";
    case CSN_E_ALIGN: return "csn: a size or leading dimension is not a multiple of 4 floats";
    case CSN_E_PTR: return "csn: device pointer not 16-byte aligned";
    case CSN_E_STRIDE: return "csn: a stride is not a multiple of 4 floats";
    case CSN_E_DIM: return "csn: unsupported head / model dimension (need 32, 64, 96, 128 or 256)";
    case CSN_E_WORKSPACE: return "csn: workspace too small";
    default: return status > 0 ? hipGetErrorString((hipError_t)status) : "csn: unknown status";
  }
}

long long csn_wgrad_workspace_floats(int rows, int cols, int n_maps, int n_points) {
  if (rows <= 0 || cols <= 0 || n_maps <= 0 || n_points <= 0) return 0;
  const int chunk = wgrad_chunk(n_maps, n_points);
  const long long n_chunks = (n_points + chunk - 1) / chunk;
  return (long long)n_maps * n_chunks * rows * cols;
}

int csn_project_f32(const float* x, long long x_shape_stride, int ld_x, const float* w, int rows, int channels,
                    float* out, long long out_shape_stride, int ld_out, int n_shapes, int n_points, int div_rows,
                    float temperature, int out_split, long long out_plane_stride, void* stream) {
  const int x16 = out_split >= 0 ? (out_split & 16) : 0;             // + 16: x is a bf16 map (math mode 2)
  if (out_split >= 0) out_split &= ~16;
  if (x16 && mode() != 2) return CSN_E_ARG;
  if (out_split && mode() == 0) return CSN_E_ARG;
  if (out_split < 0 || out_split > 3) return CSN_E_ARG;
  if (out_split == 3 && mode() < 2) return CSN_E_ARG;               // one 16-bit map: the single-product modes
  if (!x || !w || !out || rows <= 0 || channels <= 0 || n_shapes <= 0 || n_points <= 0) return CSN_E_ARG;
  if (out_split == 2) {
    // tile planes: out_plane_stride = points per attention block (<= 512), ld_out = row pitch = n_blocks * 512 * planes
    const int bp = 512 * planes_of(mode());
    if (out_plane_stride <= 0 || out_plane_stride > 512 || (out_plane_stride & 3) || (ld_out % bp)) return CSN_E_ARG;
    if (((long long)n_points + out_plane_stride - 1) / out_plane_stride * bp > ld_out) return CSN_E_ARG;
    if (out_shape_stride & 7) return CSN_E_STRIDE;
  }
  if (out_split == 1 && mode() != 1) return CSN_E_ARG;              // whole hi / lo planes: mode 1 only
  if (n_points > ld_x || (out_split != 2 && n_points > ld_out)) return CSN_E_ARG;      // a row holds the points it is read for
  if ((ld_x & 3) || (ld_out & 3) || (n_points & 3) || (channels & 3)) return CSN_E_ALIGN;
  if (mis16(x) || mis16(w) || mis16(out)) return CSN_E_PTR;
  if ((x_shape_stride & 3) || (out_shape_stride & 3)) return CSN_E_STRIDE;
  if (mode() == 1 && !x16 && (out_split == 0 || out_split == 2) && !(div_rows & 31) && csn_wx_takes(rows, channels)) {
    const int rc = launch_wx(w, x, x_shape_stride, ld_x, out, out_shape_stride, ld_out, rows, n_shapes, n_points, div_rows, temperature,
                             out_split, (int)out_plane_stride, (hipStream_t)stream);
    if (rc != CSN_NOT_TAKEN) return rc;                               // (a geometry the stream does not take: the tiled kernel below)
  }
  CsnGemmArgs g;
  g.A = operand(w, 0, 0, 0, nullptr, channels);
  g.B = operand(x, 0, 0, x_shape_stride, nullptr, ld_x);
  if (x16) g.B.fmt = CSN_FMT_16;
  g.C = operand(out, 0, 0, out_shape_stride, nullptr, ld_out);
  g.C.planes = out_split == 3 ? 1 : out_split; g.C.plane_stride = out_plane_stride;
  g.M = rows; g.N = n_points; g.K = channels;
  g.n0 = 1; g.n1 = 1; g.k_chunk = 0;
  g.alpha = 1.f; g.div_rows = div_rows; g.div_val = temperature; g.accumulate = 0; g.eval_ids = nullptr;
  return launch_gemm(g, /*b_is_nk=*/0, n_shapes, (hipStream_t)stream);
}

int csn_project_qkv_f32(const float* x, long long x_shape_stride, int ld_x, const float* w_qkv, int d_inner, int channels,
                        float* q_out, long long q_shape_stride, int ld_q, void* kv_out, long long kv_shape_stride, int ld_kv,
                        int n_shapes, int n_points, float temperature, int block, void* stream) {
  if (!q_out || !kv_out || d_inner <= 0) return CSN_E_ARG;
  if (mode() == 0) return CSN_E_ARG;                                  // tile planes: the 16-bit modes
  // one pass over x where the streaming kernel takes the product (bf16x3, 256 channels, 256 rows each of Q, K, V) ...
  if (mode() == 1 && d_inner == 256 && csn_wx_takes(3 * d_inner, channels)) {
    if (!x || !w_qkv || channels <= 0 || n_shapes <= 0 || n_points <= 0) return CSN_E_ARG;
    const int bp = 512 * planes_of(mode());
    if (block <= 0 || block > 512 || (block & 3) || (ld_kv % bp)) return CSN_E_ARG;
    if (((long long)n_points + block - 1) / block * bp > ld_kv) return CSN_E_ARG;
    if (n_points > ld_x || n_points > ld_q) return CSN_E_ARG;
    if ((ld_x & 3) || (ld_q & 3) || (n_points & 3)) return CSN_E_ALIGN;
    if (mis16(x) || mis16(w_qkv) || mis16(q_out) || mis16(kv_out)) return CSN_E_PTR;
    if ((x_shape_stride & 3) || (q_shape_stride & 3) || (kv_shape_stride & 7)) return CSN_E_STRIDE;
    CsnWxArgs a;
    a.w = w_qkv; a.x = x; a.x_item_stride = x_shape_stride; a.ldx = ld_x;
    a.out = kv_out; a.out_item_stride = kv_shape_stride; a.ldo = ld_kv;
    a.out_f32 = q_out; a.out_f32_item_stride = q_shape_stride; a.ldo_f32 = ld_q; a.n_f32 = 1;
    a.n_items = n_shapes; a.n_points = n_points; a.n_sets = 3;
    a.div_rows = d_inner; a.div_val = temperature; a.div_rcp = 1.f / temperature;
    int ex = 0;
    a.div_exact = (std::frexp(temperature, &ex) == 0.5f && temperature > 0.f) ? 1 : 0;
    a.tb = block;
    const int rc = csn_launch_wx(a, 4, (hipStream_t)stream);
    if (rc != CSN_NOT_TAKEN) return rc;
  }
  // ... two projections elsewhere (the same results: an output element is one dot product in one order on either route)
  const int rc = csn_project_f32(x, x_shape_stride, ld_x, w_qkv, d_inner, channels, q_out, q_shape_stride, ld_q, n_shapes, n_points,
                                 d_inner, temperature, 0, 0, stream);
  if (rc) return rc;
  return csn_project_f32(x, x_shape_stride, ld_x, w_qkv + (long long)d_inner * channels, 2 * d_inner, channels,
                         static_cast<float*>(kv_out), kv_shape_stride, ld_kv, n_shapes, n_points, 0, 1.f, 2, block, stream);
}

static int attn_fwd_impl(const float* q, const float* k, const float* v, long long q_shape_stride,
                         long long kv_shape_stride, const int* q_index, const int* kv_index, int ld, float* ctx,
                         long long ctx_eval_stride, float* scores, float* lse, int n_evals, int n_heads,
                         int d_head, int block, int n_blocks, int score_pitch, float rescale_threshold,
                         float dropout_p, unsigned long long seed, int qkv_split, long long qkv_plane_stride,
                         int block_q, int ld_kv, void* stream, const int* tq_arr = nullptr, const int* t_arr = nullptr,
                         const int* eval_ids = nullptr, const int* group_offsets = nullptr, int n_groups = 0) {
  // block_q / ld_kv != 0: the queries of a block are counted separately from its keys (block = keys) and K/V maps have
  // their own leading dimension; key counts need not be multiples of 4 then (fp32 K/V maps only).  tq_arr / t_arr: the
  // per-evaluation counts of a ragged batch (block_q / block are then the maxima)
  const int bq = block_q > 0 ? block_q : block, lk = ld_kv > 0 ? ld_kv : ld;
  if (block_q < 0 || ld_kv < 0 || (ld_kv & 3)) return CSN_E_ARG;
  if (n_blocks <= 0 || block <= 0) return CSN_E_ARG;
  const int t_last = last_block_points(block, n_blocks, ld, block_q);
  if (t_last < 0) return t_last;
  if (block_q != 0 && (long long)n_blocks * ((block + 3) / 4 * 4) > lk) return CSN_E_ARG;
  if ((block & 3) && (block_q == 0 || qkv_split)) return CSN_E_ALIGN;
  if (dropout_p < 0.f || dropout_p >= 1.f) return CSN_E_ARG;
  if (qkv_split && mode() == 0) return CSN_E_ARG;
  if (!qkv_split && mode() >= 2) return CSN_E_ARG;                   // single-product modes take K / V as tile planes
  const int bp = 512 * planes_of(mode());
  if (qkv_split && (qkv_plane_stride <= 0 || (qkv_plane_stride % bp) || qkv_plane_stride < (long long)n_blocks * bp ||
                    block > 512 || (kv_shape_stride & 7)))
    return CSN_E_ARG;
  if (!q || !k || !v || !ctx || n_evals <= 0 || n_heads <= 0 || block <= 0 || n_blocks <= 0) return CSN_E_ARG;
  if (!dim_ok(d_head)) return CSN_E_DIM;
  if ((ld & 3) || (score_pitch & 3) || score_pitch < (block + 3) / 4 * 4) return CSN_E_ALIGN;
  if (mis16(q) || mis16(k) || mis16(v) || mis16(ctx) || mis16(scores)) return CSN_E_PTR;
  if ((q_shape_stride & 3) || (kv_shape_stride & 3) || (ctx_eval_stride & 3)) return CSN_E_STRIDE;
  if (block_q != 0 && (long long)n_blocks * bq > ld) return CSN_E_ARG;
  CsnAttnArgs a;
  a.T_last = t_last;
  a.Tq = block_q; a.ld_kv = ld_kv;
  a.q = q; a.k = k; a.v = v;
  a.q_shape_stride = q_shape_stride; a.kv_shape_stride = kv_shape_stride;
  a.kv_ld = (int)qkv_plane_stride;
  a.q_index = q_index; a.kv_index = kv_index; a.ld = ld;
  a.out = ctx; a.out_eval_stride = ctx_eval_stride;
  a.scores = scores; a.dscores = nullptr; a.lse = lse; a.delta = nullptr; a.ctx = nullptr;
  a.E = n_evals; a.H = n_heads; a.T = block; a.Tp = score_pitch; a.n_blocks = n_blocks;
  a.rescale_threshold = rescale_threshold;
  a.eval_ids = eval_ids; a.grp_off = nullptr; a.out_index = nullptr; a.accumulate = 0;
  if (group_offsets) {                                               // grouped by query slot (16-bit modes; fp32: the listed evaluations, ungrouped)
    if (!eval_ids || n_groups <= 0 || n_groups > n_evals || block_q != 0 || tq_arr || t_arr) return CSN_E_ARG;
    if (mode() != 0) { a.grp_off = group_offsets; a.E = n_groups; }
  }
  a.dropout_p = dropout_p; a.seed = seed;
  a.r_planes = 0; a.kv_planes = qkv_split; a.r_plane_stride = 0; a.kv_plane_stride = 0; a.sc_tiles = 0;
  a.tq_arr = tq_arr; a.t_arr = t_arr;
  if (act16()) {                                                     // Qs in, Ctx out: 16-bit maps of the mode's type
    if (!act16_fwd_ok() || !qkv_split || block_q != 0) return CSN_E_ARG;
    a.r_fmt = a.out_fmt = act16();
  }
  if (t_score_layout && scores) {                                    // tile-major scores: block mode, bf16x3, tile-plane K / V
    if (mode() != 1 || !qkv_split || block_q != 0 || tq_arr || t_arr || score_pitch < (block + 31) / 32 * 32) return CSN_E_ARG;
    a.sc_layout = 1;
  }
  return mode() != 0 ? csn_launch_attn_fwd_bf16x3(a, d_head, mode(), (hipStream_t)stream)
                     : csn_launch_attn_fwd_f32(a, d_head, (hipStream_t)stream);
}

int csn_block_attn_fwd_f32(const float* q, const float* k, const float* v, long long q_shape_stride,
                           long long kv_shape_stride, const int* q_index, const int* kv_index, int ld, float* ctx,
                           long long ctx_eval_stride, float* scores, float* lse, int n_evals, int n_heads,
                           int d_head, int block, int n_blocks, int score_pitch, float rescale_threshold,
                           float dropout_p, unsigned long long seed, int qkv_split, long long qkv_plane_stride,
                           void* stream) {
  return attn_fwd_impl(q, k, v, q_shape_stride, kv_shape_stride, q_index, kv_index, ld, ctx, ctx_eval_stride, scores, lse,
                       n_evals, n_heads, d_head, block, n_blocks, score_pitch, rescale_threshold, dropout_p, seed, qkv_split,
                       qkv_plane_stride, 0, 0, stream);
}

int csn_block_attn_fwd_grouped_f32(const float* q, const float* k, const float* v, long long q_shape_stride,
                                   long long kv_shape_stride, const int* q_index, const int* kv_index, int ld, float* ctx,
                                   long long ctx_eval_stride, float* scores, float* lse, int n_evals, int n_heads,
                                   int d_head, int block, int n_blocks, int score_pitch, float rescale_threshold,
                                   float dropout_p, unsigned long long seed, int qkv_split, long long qkv_plane_stride,
                                   const int* eval_ids, const int* group_offsets, int n_groups, void* stream) {
  if (!eval_ids || !group_offsets) return CSN_E_ARG;
  return attn_fwd_impl(q, k, v, q_shape_stride, kv_shape_stride, q_index, kv_index, ld, ctx, ctx_eval_stride, scores, lse,
                       n_evals, n_heads, d_head, block, n_blocks, score_pitch, rescale_threshold, dropout_p, seed, qkv_split,
                       qkv_plane_stride, 0, 0, stream, nullptr, nullptr, eval_ids, group_offsets, n_groups);
}

int csn_cross_attn_fwd_f32(const float* q, const float* k, const float* v, long long q_shape_stride,
                           long long kv_shape_stride, int ld_q, int ld_kv, float* ctx, long long ctx_eval_stride,
                           float* scores, float* lse, int n_evals, int n_heads, int d_head, int n_queries, int n_keys,
                           int score_pitch, float rescale_threshold, float dropout_p, unsigned long long seed, void* stream) {
  if (n_queries <= 0 || n_keys <= 0) return CSN_E_ARG;
  if (n_queries & 3) return CSN_E_ALIGN;
  ModeGuard guard(mode() >= 2 ? 1 : mode());
  return attn_fwd_impl(q, k, v, q_shape_stride, kv_shape_stride, nullptr, nullptr, ld_q, ctx, ctx_eval_stride, scores, lse,
                       n_evals, n_heads, d_head, n_keys, 1, score_pitch, rescale_threshold, dropout_p, seed, 0, 0, n_queries,
                       ld_kv, stream);
}

static int attn_bwd_dq_impl(const float* dctx, const float* ctx, long long ctx_eval_stride, const float* k,
                            const float* v, long long kv_shape_stride, const int* kv_index, int ld, float* scores,
                            float* dscores, const float* lse, float* delta, float* dq, long long dq_slot_stride,
                            const int* dq_index, int accumulate, const int* eval_ids, int n_launch_evals, int n_heads,
                            int d_head, int block, int n_blocks, int score_pitch, float dropout_p,
                            unsigned long long seed, int dctx_split, long long dctx_plane_stride, int kv_split,
                            long long kv_plane_stride, int probs_tiles, int block_q, int ld_kv, const int* group_offsets,
                            int n_groups, void* stream, const int* tq_arr = nullptr, const int* t_arr = nullptr,
                            const float* q = nullptr, long long q_shape_stride = 0, const int* q_index = nullptr) {
  if (block_q < 0 || ld_kv < 0 || (ld_kv & 3)) return CSN_E_ARG;
  // q != NULL: the scores are recomputed from the pre-scaled queries (block mode, tile-plane K / V, where the kernel has room)
  if (q && (block_q != 0 || !kv_split || !(csn_attn_bwd_grouping(d_head, block) & 4) || mis16(q))) return CSN_E_ARG;
  if (q && (q_shape_stride & 3)) return CSN_E_STRIDE;
  if ((tq_arr || t_arr) && (group_offsets || n_blocks != 1)) return CSN_E_ARG;
  if (n_blocks <= 0 || block <= 0) return CSN_E_ARG;
  const int t_last = last_block_points(block, n_blocks, ld, block_q);
  if (t_last < 0) return t_last;
  if (block_q != 0 && ((long long)n_blocks * block_q > ld || (long long)n_blocks * ((block + 3) / 4 * 4) > (ld_kv > 0 ? ld_kv : ld)))
    return CSN_E_ARG;                                                 // cross-length: queries and keys fit their rows
  if (group_offsets && (n_groups <= 0 || !eval_ids || !(csn_attn_bwd_grouping(d_head, block) & 1))) return CSN_E_ARG;
  if ((block & 3) && (block_q == 0 || kv_split)) return CSN_E_ALIGN;
  if (probs_tiles && (mode() == 0 || score_pitch < (block + 31) / 32 * 32)) return CSN_E_ARG;
  if (dropout_p < 0.f || dropout_p >= 1.f) return CSN_E_ARG;
  if (dctx_split) return CSN_E_ARG;                                  // reserved (see header)
  if (kv_split && mode() == 0) return CSN_E_ARG;
  if (kv_split < 0 || kv_split > 2 || (kv_split == 2 && mode() != 2)) return CSN_E_ARG;   // 2: fp16 planes of a mode-3 forward
  if (mode() == 3) return CSN_E_ARG;                                 // fp16: forward only — run the backward in mode 2
  if (mode() == 2 && !(kv_split && (probs_tiles || q))) return CSN_E_ARG;   // single-product mode: tile planes in and out
  const int bp = 512 * planes_of(mode());
  if (kv_split && (kv_plane_stride <= 0 || (kv_plane_stride % bp) || kv_plane_stride < (long long)n_blocks * bp ||
                   block > 512 || (kv_shape_stride & 7)))
    return CSN_E_ARG;
  // (recomputed scores with probs_tiles == 0: nothing is read from or written to scores / dscores — they may be NULL)
  // (recomputed scores in the one-plane mode: P and dS both travel in dscores, `scores` is never touched)
  const bool need_sc = !q || (probs_tiles && mode() == 1), need_ds = !q || probs_tiles;
  if (!dctx || !ctx || !k || !v || !lse || !delta || !dq || (need_sc && !scores) || (need_ds && !dscores)) return CSN_E_ARG;
  if (n_launch_evals <= 0 || n_heads <= 0 || block <= 0 || n_blocks <= 0) return CSN_E_ARG;
  if (!dim_ok(d_head)) return CSN_E_DIM;
  if ((ld & 3) || (score_pitch & 3) || score_pitch < (block + 3) / 4 * 4) return CSN_E_ALIGN;
  if (mis16(dctx) || mis16(k) || mis16(v) || mis16(scores) || mis16(dscores) || mis16(dq)) return CSN_E_PTR;
  if ((kv_shape_stride & 3) || (ctx_eval_stride & 3) || (dq_slot_stride & 3)) return CSN_E_STRIDE;
  hipStream_t st = (hipStream_t)stream;
  // delta[e][h][n] = sum_c dctx * ctx (softmax backward row constant) is formed in the kernel's prologue, where
  // the dctx columns are being loaded anyway
  CsnAttnArgs a;
  a.ctx = ctx;
  a.T_last = t_last;
  a.Tq = block_q; a.ld_kv = ld_kv;
  a.q = dctx; a.k = k; a.v = v;
  a.q_shape_stride = dctx_split ? 2 * ctx_eval_stride : ctx_eval_stride;     // split dctx: [eval][2 planes][D][ld]
  a.kv_shape_stride = kv_shape_stride;
  a.q_index = nullptr; a.kv_index = kv_index; a.ld = ld;
  a.out = dq; a.out_eval_stride = dq_slot_stride;
  a.scores = scores; a.dscores = dscores; a.lse = const_cast<float*>(lse); a.delta = delta;
  a.E = group_offsets ? n_groups : n_launch_evals; a.H = n_heads; a.T = block; a.Tp = score_pitch; a.n_blocks = n_blocks;
  a.rescale_threshold = 0.f;
  a.eval_ids = eval_ids; a.grp_off = group_offsets; a.out_index = dq_index; a.accumulate = accumulate;
  a.dropout_p = dropout_p; a.seed = seed;
  a.r_planes = 0; a.kv_planes = kv_split != 0; a.r_plane_stride = 0; a.kv_plane_stride = 0; a.kv_ld = (int)kv_plane_stride;
  a.kv_f16 = kv_split == 2;
  a.sc_tiles = probs_tiles;
  a.tq_arr = tq_arr; a.t_arr = t_arr;
  a.q2 = q; a.q2_shape_stride = q_shape_stride; a.q2_index = q_index;
  if (t_score_layout) {                                              // tile-major scores in, tile-major P / dS planes out
    if (mode() != 1 || !kv_split || block_q != 0 || tq_arr || t_arr || !probs_tiles) return CSN_E_ARG;
    a.sc_layout = 1;
  }
  if (act16()) {                                                     // dO: bf16; O and Qs: the forward's 16-bit type; dQ stays fp32
    if (!act16_bwd_ok() || !kv_split || block_q != 0) return CSN_E_ARG;
    a.r_fmt = 1; a.ctx_fmt = act16(); a.q2_fmt = act16();
    if (grad16()) {
      if (accumulate) return CSN_E_ARG;                               // a 16-bit gradient map is written once (grouped calls)
      a.out_fmt = 1;
    }
  }
  return mode() != 0 ? csn_launch_attn_bwd_bf16x3(a, d_head, mode(), st) : csn_launch_attn_bwd_f32(a, d_head, st);
}

int csn_block_attn_bwd_dq_f32(const float* dctx, const float* ctx, long long ctx_eval_stride, const float* k,
                              const float* v, long long kv_shape_stride, const int* kv_index, int ld, float* scores,
                              float* dscores, const float* lse, float* delta, float* dq, long long dq_slot_stride,
                              const int* dq_index, int accumulate, const int* eval_ids, int n_launch_evals, int n_heads,
                              int d_head, int block, int n_blocks, int score_pitch, float dropout_p,
                              unsigned long long seed, int dctx_split, long long dctx_plane_stride, int kv_split,
                              long long kv_plane_stride, int probs_tiles, const int* group_offsets, int n_groups,
                              void* stream) {
  return attn_bwd_dq_impl(dctx, ctx, ctx_eval_stride, k, v, kv_shape_stride, kv_index, ld, scores, dscores, lse, delta, dq,
                          dq_slot_stride, dq_index, accumulate, eval_ids, n_launch_evals, n_heads, d_head, block, n_blocks,
                          score_pitch, dropout_p, seed, dctx_split, dctx_plane_stride, kv_split, kv_plane_stride, probs_tiles,
                          0, 0, group_offsets, n_groups, stream);
}

int csn_block_attn_bwd_dq_recompute_f32(const float* dctx, const float* ctx, long long ctx_eval_stride, const float* q,
                                        long long q_shape_stride, const int* q_index, const float* k, const float* v,
                                        long long kv_shape_stride, const int* kv_index, int ld, float* probs,
                                        float* dscores, const float* lse, float* delta, float* dq,
                                        long long dq_slot_stride, const int* dq_index, int accumulate, const int* eval_ids,
                                        int n_launch_evals, int n_heads, int d_head, int block, int n_blocks,
                                        int score_pitch, float dropout_p, unsigned long long seed,
                                        long long kv_plane_stride, int kv_f16, int probs_tiles, const int* group_offsets,
                                        int n_groups, void* stream) {
  if (!q) return CSN_E_ARG;
  return attn_bwd_dq_impl(dctx, ctx, ctx_eval_stride, k, v, kv_shape_stride, kv_index, ld, probs, dscores, lse, delta, dq,
                          dq_slot_stride, dq_index, accumulate, eval_ids, n_launch_evals, n_heads, d_head, block, n_blocks,
                          score_pitch, dropout_p, seed, 0, 0, kv_f16 ? 2 : 1, kv_plane_stride, probs_tiles, 0, 0, group_offsets,
                          n_groups, stream, nullptr, nullptr, q, q_shape_stride, q_index);
}

static int attn_bwd_dkv_impl(const float* dctx, long long ctx_eval_stride, const float* q, long long q_shape_stride,
                             const int* q_index, int ld, const float* probs, const float* dscores, float* dk,
                             float* dv, long long dkv_slot_stride, const int* dk_index, const int* dv_index,
                             int accumulate, const int* eval_ids, int n_launch_evals, int n_heads, int d_head,
                             int block, int n_blocks, int score_pitch, int dctx_split, long long dctx_plane_stride,
                             int q_split, long long q_plane_stride, int probs_tiles, int block_q, int ld_kv,
                             const int* group_offsets, int n_groups, void* stream, const int* tq_arr = nullptr,
                             const int* t_arr = nullptr) {
  // block_q / ld_kv != 0 (cross-length attention): block counts the keys, block_q (% 4) the queries that are contracted;
  // dk / dv are [d][ld_kv] maps whose columns block .. round-up-4(block) are written as zeros
  if (block_q < 0 || ld_kv < 0 || (ld_kv & 3) || (block_q & 3)) return CSN_E_ARG;
  if ((block & 3) && block_q == 0) return CSN_E_ALIGN;
  const int bq = block_q > 0 ? block_q : block, lk = ld_kv > 0 ? ld_kv : ld, bk4 = (block + 3) / 4 * 4;
  if (n_blocks <= 0 || block <= 0) return CSN_E_ARG;
  const int t_last = last_block_points(block, n_blocks, ld, block_q);
  if (t_last < 0) return t_last;
  if (block_q != 0 && ((long long)n_blocks * bk4 > lk || (long long)n_blocks * bq > ld)) return CSN_E_ARG;
  if (probs_tiles && (mode() == 0 || score_pitch < (block + 31) / 32 * 32)) return CSN_E_ARG;
  if (mode() == 3 || (mode() == 2 && !probs_tiles)) return CSN_E_ARG;
  if (dctx_split || q_split) return CSN_E_ARG;                       // reserved (see header)
  if (group_offsets && (n_groups <= 0 || !eval_ids || !(csn_attn_bwd_grouping(d_head, block) & 2))) return CSN_E_ARG;
  if (!dctx || !q || !dscores || !dk || !dv) return CSN_E_ARG;
  if (!probs && !(probs_tiles && mode() == 2)) return CSN_E_ARG;       // (one plane: P and dS are both read from dscores)
  if (n_launch_evals <= 0 || n_heads <= 0 || block <= 0 || n_blocks <= 0) return CSN_E_ARG;
  if (!dim_ok(d_head)) return CSN_E_DIM;
  if ((ld & 3) || (score_pitch & 3) || score_pitch < bk4) return CSN_E_ALIGN;
  if (mis16(dctx) || mis16(q) || mis16(probs) || mis16(dscores) || mis16(dk) || mis16(dv)) return CSN_E_PTR;
  if ((q_shape_stride & 3) || (ctx_eval_stride & 3) || (dkv_slot_stride & 3)) return CSN_E_STRIDE;
  hipStream_t st = (hipStream_t)stream;
  // dV^T[c][key] (+)= sum_q dO^T[c][q] P[q][key]   and   dK^T[d][key] (+)= sum_q Qs^T[d][q] dS[q][key]
  // (the score blocks are stored [query][key]: k-major B operands)
  const long long blk_sc = (long long)bq * score_pitch;
  CsnGemmArgs g;
  g.M = d_head; g.N = bk4; g.K = bq;
  g.n0 = n_blocks; g.n1 = n_heads; g.k_chunk = 0;
  g.alpha = 1.f; g.div_rows = 0; g.div_val = 1.f; g.accumulate = accumulate; g.eval_ids = eval_ids;
  if (group_offsets) { g.eval_ids = nullptr; g.grp_off = group_offsets; g.grp_items = eval_ids; }
  g.n_arr = t_arr; g.k_arr = tq_arr;            // ragged batch: keys (rounded up to 4 by the kernel) / queries of every evaluation
  g.n_last = g.k_last = t_last;                 // the row ends inside the last block
  const int n_batch = group_offsets ? n_groups : n_launch_evals;
  int rc = 0;
  if (act16() && (!act16_bwd_ok() || !probs_tiles || block_q != 0)) return CSN_E_ARG;
  g.A = operand(dctx, bq, (long long)d_head * ld, dctx_split ? 2 * ctx_eval_stride : ctx_eval_stride, nullptr, ld);
  g.A.planes = dctx_split; g.A.plane_stride = dctx_plane_stride;
  if (act16()) g.A.fmt = CSN_FMT_16;                                 // dO: a bf16 map
  // tile planes: the same buffers viewed as 16-bit elements (two per float: block strides double).  Two planes (mode 1): a
  // row of 16 tiles [hi | lo] is the fp32 row, pitch 2 * score_pitch, P in `probs`, dS in `dscores`.  One plane (mode 2):
  // compact rows of pitch score_pitch, both in `dscores` — per block [P: bq rows | dS: bq rows] (`probs` is not read)
  const int bm = probs_tiles ? 2 : 1;
  const bool one_plane = probs_tiles && mode() == 2;
  const int ldb = one_plane ? score_pitch : bm * score_pitch;
  const float* p_src = one_plane ? dscores : probs;
  const float* ds_src = one_plane ? reinterpret_cast<const float*>(reinterpret_cast<const short*>(dscores) + blk_sc) : dscores;
  // tile-major P / dS planes (csn_set_thread_score_layout): where the dV / dK products run on the 16-wave 256 x 256 kernel
  const bool tile_major = t_score_layout != 0;
  if (tile_major && (mode() != 1 || !probs_tiles || block_q != 0 || tq_arr || t_arr || !csn_gemm_tile_major_planes(d_head, bk4)))
    return CSN_E_ARG;
  g.B = operand(p_src, bm * blk_sc, bm * blk_sc * n_blocks, bm * blk_sc * n_blocks * n_heads, nullptr, ldb);
  g.B.planes = probs_tiles ? (tile_major ? 3 : 2) : 0;
  const bool g16 = act16() && grad16();
  if (g16 && accumulate) return CSN_E_ARG;                            // a 16-bit gradient map is written once (grouped calls)
  g.C = operand(dv, block, (long long)d_head * lk, dkv_slot_stride, dv_index, lk);
  if (g16) g.C.planes = 1;
  rc = launch_gemm(g, 0, n_blocks * n_heads * n_batch, st);
  if (rc) return rc;
  g.A = operand(q, bq, (long long)d_head * ld, q_shape_stride, q_index, ld);
  g.A.planes = q_split; g.A.plane_stride = q_plane_stride;
  if (act16()) g.A.fmt = act16() == 2 ? CSN_FMT_F16_TO_BF16 : CSN_FMT_16;   // Qs: the forward's 16-bit map
  g.B = operand(ds_src, bm * blk_sc, bm * blk_sc * n_blocks, bm * blk_sc * n_blocks * n_heads, nullptr, ldb);
  g.B.planes = probs_tiles ? (tile_major ? 3 : 2) : 0;
  g.C = operand(dk, block, (long long)d_head * lk, dkv_slot_stride, dk_index, lk);
  if (g16) g.C.planes = 1;
  return launch_gemm(g, 0, n_blocks * n_heads * n_batch, st);
}

int csn_block_attn_bwd_dkv_f32(const float* dctx, long long ctx_eval_stride, const float* q, long long q_shape_stride,
                               const int* q_index, int ld, const float* probs, const float* dscores, float* dk,
                               float* dv, long long dkv_slot_stride, const int* dk_index, const int* dv_index,
                               int accumulate, const int* eval_ids, int n_launch_evals, int n_heads, int d_head,
                               int block, int n_blocks, int score_pitch, int dctx_split, long long dctx_plane_stride,
                               int q_split, long long q_plane_stride, int probs_tiles, const int* group_offsets,
                               int n_groups, void* stream) {
  return attn_bwd_dkv_impl(dctx, ctx_eval_stride, q, q_shape_stride, q_index, ld, probs, dscores, dk, dv, dkv_slot_stride,
                           dk_index, dv_index, accumulate, eval_ids, n_launch_evals, n_heads, d_head, block, n_blocks,
                           score_pitch, dctx_split, dctx_plane_stride, q_split, q_plane_stride, probs_tiles, 0, 0,
                           group_offsets, n_groups, stream);
}

int csn_attn_bwd_grouping(int d_head, int block) {
  if (mode() == 0) return 0;
  const bool tiles_ok = mode() != 3 && block <= 512 && !(block & 3) && dim_ok(d_head);
  const bool recompute = tiles_ok && csn_attn_recompute_fits(planes_of(mode()), d_head / 32);
  const bool flash = recompute && csn_attn_dkv_flash_fits(d_head / 32);
  const bool tm = mode() == 1 && tiles_ok && csn_gemm_tile_major_planes(d_head, (block + 3) / 4 * 4);
  return 1 | (csn_gemm_bf16x3_big_tiles(d_head, (block + 3) / 4 * 4) ? 2 : 0) | (recompute ? 4 : 0) | (flash ? 8 : 0) | (tm ? 16 : 0);
}

int csn_block_attn_bwd_dkv_flash_f32(const float* dctx, long long ctx_eval_stride, const float* q, long long q_shape_stride,
                                     const int* q_index, const float* k, const float* v, long long kv_shape_stride,
                                     const int* kv_index, long long kv_plane_stride, int kv_f16, int ld, const float* lse,
                                     const float* delta, float* dk, float* dv, long long dkv_slot_stride,
                                     const int* dk_index, const int* dv_index, int accumulate, const int* eval_ids,
                                     int n_launch_evals, int n_heads, int d_head, int block, int n_blocks, int score_pitch,
                                     float dropout_p, unsigned long long seed, const int* group_offsets, int n_groups,
                                     void* stream) {
  if (!dctx || !q || !k || !v || !lse || !delta || !dk || !dv) return CSN_E_ARG;
  if (n_launch_evals <= 0 || n_heads <= 0 || block <= 0 || n_blocks <= 0) return CSN_E_ARG;
  if (!dim_ok(d_head)) return CSN_E_DIM;
  if (!(csn_attn_bwd_grouping(d_head, block) & 8)) return CSN_E_ARG;
  if (group_offsets && (n_groups <= 0 || !eval_ids)) return CSN_E_ARG;
  if (dropout_p < 0.f || dropout_p >= 1.f) return CSN_E_ARG;
  if (kv_f16 && mode() != 2) return CSN_E_ARG;
  const int t_last = last_block_points(block, n_blocks, ld, 0);
  if (t_last < 0) return t_last;
  const int bp = 512 * planes_of(mode());
  if (kv_plane_stride <= 0 || (kv_plane_stride % bp) || kv_plane_stride < (long long)n_blocks * bp || (kv_shape_stride & 7))
    return CSN_E_ARG;
  if ((ld & 3) || (score_pitch & 3) || score_pitch < block) return CSN_E_ALIGN;
  if (mis16(dctx) || mis16(q) || mis16(k) || mis16(v) || mis16(dk) || mis16(dv)) return CSN_E_PTR;
  if ((q_shape_stride & 3) || (ctx_eval_stride & 3) || (dkv_slot_stride & 3)) return CSN_E_STRIDE;
  CsnAttnDkvArgs a;
  a.q = q; a.q_shape_stride = q_shape_stride; a.q_index = q_index;
  a.dctx = dctx; a.ctx_eval_stride = ctx_eval_stride;
  a.k = k; a.v = v; a.kv_shape_stride = kv_shape_stride; a.kv_ld = (int)kv_plane_stride; a.kv_index = kv_index;
  a.lse = lse; a.delta = delta;
  a.dk = dk; a.dv = dv; a.dkv_slot_stride = dkv_slot_stride; a.dk_index = dk_index; a.dv_index = dv_index;
  a.accumulate = accumulate;
  a.eval_ids = eval_ids; a.grp_off = group_offsets; a.n_groups = group_offsets ? n_groups : n_launch_evals;
  a.ld = ld; a.H = n_heads; a.T = block; a.Tp = score_pitch; a.n_blocks = n_blocks; a.T_last = t_last;
  a.dropout_p = dropout_p; a.seed = seed; a.kv_f16 = kv_f16 != 0;
  if (act16()) {
    if (!act16_bwd_ok()) return CSN_E_ARG;
    a.q_fmt = act16(); a.dctx_fmt = 1;
    if (grad16()) {
      if (accumulate) return CSN_E_ARG;
      a.out_fmt = 1;
    }
  }
  return csn_launch_attn_dkv_flash(a, d_head, mode(), (hipStream_t)stream);
}

/* cross-length attention backward (MinkowskiNet/models/attention.py: one unchunked block per evaluation, n_queries != n_keys;
 * gradients flow to queries, keys and values).  n_queries % 4 == 0 (pad with zero points); n_keys arbitrary. */
int csn_cross_attn_bwd_f32(const float* dctx, const float* ctx, long long ctx_eval_stride, const float* q, const float* k,
                           const float* v, long long q_shape_stride, long long kv_shape_stride, int ld_q, int ld_kv,
                           float* scores, float* dscores, const float* lse, float* delta, float* dq, float* dk, float* dv,
                           long long dq_eval_stride, long long dkv_eval_stride, int n_evals, int n_heads, int d_head,
                           int n_queries, int n_keys, int score_pitch, float dropout_p, unsigned long long seed,
                           void* stream) {
  if (n_queries <= 0 || n_keys <= 0) return CSN_E_ARG;
  if (n_queries & 3) return CSN_E_ALIGN;
  ModeGuard guard(mode() >= 2 ? 1 : mode());
  const int pt = (mode() != 0 && score_pitch >= (n_keys + 31) / 32 * 32) ? 1 : 0;
  int rc = attn_bwd_dq_impl(dctx, ctx, ctx_eval_stride, k, v, kv_shape_stride, nullptr, ld_q, scores, dscores, lse, delta, dq,
                            dq_eval_stride, nullptr, 0, nullptr, n_evals, n_heads, d_head, n_keys, 1, score_pitch, dropout_p,
                            seed, 0, 0, 0, 0, pt, n_queries, ld_kv, nullptr, 0, stream);
  if (rc) return rc;
  return attn_bwd_dkv_impl(dctx, ctx_eval_stride, q, q_shape_stride, nullptr, ld_q, scores, dscores, dk, dv, dkv_eval_stride,
                           nullptr, nullptr, 0, nullptr, n_evals, n_heads, d_head, n_keys, 1, score_pitch, 0, 0, 0, 0, pt,
                           n_queries, ld_kv, nullptr, 0, stream);
}

/* ragged batches of the cross-length attention: the same kernels with per-evaluation query / key counts (device arrays) */
int csn_varlen_attn_fwd_f32(const float* q, const float* k, const float* v, long long q_shape_stride,
                            long long kv_shape_stride, int ld_q, int ld_kv, float* ctx, long long ctx_eval_stride,
                            float* scores, float* lse, int n_evals, int n_heads, int d_head, int max_queries, int max_keys,
                            const int* n_queries, const int* n_keys, int score_pitch, float rescale_threshold,
                            float dropout_p, unsigned long long seed, void* stream) {
  if (max_queries <= 0 || max_keys <= 0 || !n_queries || !n_keys) return CSN_E_ARG;
  if (max_queries & 3) return CSN_E_ALIGN;
  ModeGuard guard(mode() >= 2 ? 1 : mode());
  return attn_fwd_impl(q, k, v, q_shape_stride, kv_shape_stride, nullptr, nullptr, ld_q, ctx, ctx_eval_stride, scores, lse,
                       n_evals, n_heads, d_head, max_keys, 1, score_pitch, rescale_threshold, dropout_p, seed, 0, 0, max_queries,
                       ld_kv, stream, n_queries, n_keys);
}

int csn_varlen_attn_bwd_f32(const float* dctx, const float* ctx, long long ctx_eval_stride, const float* q, const float* k,
                            const float* v, long long q_shape_stride, long long kv_shape_stride, int ld_q, int ld_kv,
                            float* scores, float* dscores, const float* lse, float* delta, float* dq, float* dk, float* dv,
                            long long dq_eval_stride, long long dkv_eval_stride, int n_evals, int n_heads, int d_head,
                            int max_queries, int max_keys, const int* n_queries, const int* n_keys, int score_pitch,
                            float dropout_p, unsigned long long seed, void* stream) {
  if (max_queries <= 0 || max_keys <= 0 || !n_queries || !n_keys) return CSN_E_ARG;
  if (max_queries & 3) return CSN_E_ALIGN;
  ModeGuard guard(mode() >= 2 ? 1 : mode());
  const int pt = (mode() != 0 && score_pitch >= (max_keys + 31) / 32 * 32) ? 1 : 0;
  int rc = attn_bwd_dq_impl(dctx, ctx, ctx_eval_stride, k, v, kv_shape_stride, nullptr, ld_q, scores, dscores, lse, delta, dq,
                            dq_eval_stride, nullptr, 0, nullptr, n_evals, n_heads, d_head, max_keys, 1, score_pitch, dropout_p,
                            seed, 0, 0, 0, 0, pt, max_queries, ld_kv, nullptr, 0, stream, n_queries, n_keys);
  if (rc) return rc;
  return attn_bwd_dkv_impl(dctx, ctx_eval_stride, q, q_shape_stride, nullptr, ld_q, scores, dscores, dk, dv, dkv_eval_stride,
                           nullptr, nullptr, 0, nullptr, n_evals, n_heads, d_head, max_keys, 1, score_pitch, 0, 0, 0, 0, pt,
                           max_queries, ld_kv, nullptr, 0, stream, n_queries, n_keys);
}

long long csn_masked_ce_workspace_bytes(int n_shapes, int n_points) {
  if (n_shapes <= 0 || n_points <= 0) return 0;
  return csn_masked_ce_blocks(n_shapes, n_points) * 3 * (long long)sizeof(double);
}

int csn_masked_ce_fwd_f32(const float* logits, long long shape_stride, int ld, const long long* labels, long long label_shape_stride,
                          int n_shapes, int n_classes, int n_points, int mask, float* lse, void* ws, long long ws_bytes,
                          float* stats, void* stream) {
  if (!logits || !labels || !lse || !ws || !stats || n_shapes <= 0 || n_classes <= 0 || n_points <= 0 || n_points > ld) return CSN_E_ARG;
  if (n_shapes > 65535 || n_classes > 65535) return CSN_E_DIM;
  if (reinterpret_cast<unsigned long long>(ws) & 7) return CSN_E_PTR;
  if (ws_bytes < csn_masked_ce_workspace_bytes(n_shapes, n_points)) return CSN_E_WORKSPACE;
  CsnMaskedCeArgs a{};
  a.logits = logits; a.shape_stride = shape_stride; a.ld = ld; a.labels = labels; a.label_shape_stride = label_shape_stride;
  a.n_shapes = n_shapes; a.n_classes = n_classes; a.n_points = n_points; a.mask = mask;
  a.lse = lse; a.partials = static_cast<double*>(ws); a.stats = stats;
  return csn_launch_masked_ce_fwd(a, (hipStream_t)stream);
}

int csn_masked_ce_bwd_f32(const float* logits, long long shape_stride, int ld, const long long* labels, long long label_shape_stride,
                          int n_shapes, int n_classes, int n_points, int mask, const float* lse, const float* stats,
                          const float* grad_out, float* dlogits, long long dshape_stride, int dld, void* stream) {
  if (!logits || !labels || !lse || !stats || !grad_out || !dlogits || n_shapes <= 0 || n_classes <= 0 || n_points <= 0 || n_points > ld ||
      n_points > dld)
    return CSN_E_ARG;
  if (n_shapes > 65535 || n_classes > 65535) return CSN_E_DIM;
  // (any point count, pitch and alignment, like the forward: the launcher takes the 16-byte form where the geometry allows it)
  CsnMaskedCeArgs a{};
  a.logits = logits; a.shape_stride = shape_stride; a.ld = ld; a.labels = labels; a.label_shape_stride = label_shape_stride;
  a.n_shapes = n_shapes; a.n_classes = n_classes; a.n_points = n_points; a.mask = mask;
  a.lse = const_cast<float*>(lse); a.stats = const_cast<float*>(stats); a.grad_out = grad_out;
  a.dlogits = dlogits; a.dshape_stride = dshape_stride; a.dld = dld;
  return csn_launch_masked_ce_bwd(a, (hipStream_t)stream);
}

long long csn_outproj_ln_workspace_floats(int n_evals, int d_model, int d_inner, int n_points) {
  if (n_evals <= 0 || d_model <= 0 || d_inner <= 0 || n_points <= 0) return 0;
  const long long tiled = (long long)n_evals * ((n_points + 255) / 256) * d_model;
  const long long stream = d_model == 256 && d_inner == 256 ? (long long)n_evals * csn_wx_ln_sum_slots(n_evals, n_points) * 256 : 0;
  return tiled > stream ? tiled : stream;
}

int csn_outproj_ln_fwd_f32(const float* ctx, long long ctx_eval_stride, const float* wfc, const float* xres,
                           long long xres_shape_stride, const int* res_index, float* xhat,
                           long long xhat_eval_stride, float* rstd, int n_evals, int d_model, int d_inner, int ld,
                           int n_points, float eps, float dropout_p, unsigned long long seed, float* xhat_sum,
                           float* sum_ws, long long sum_ws_floats, void* stream) {
  if (dropout_p < 0.f || dropout_p >= 1.f) return CSN_E_ARG;
  if (!ctx || !wfc || !xres || !xhat || !rstd || n_evals <= 0 || n_points <= 0 || d_inner <= 0 || n_points > ld) return CSN_E_ARG;
  if (dropout_p > 0.f && (long long)(d_model / 2 + 1) * ld >= (1ll << 32)) return CSN_E_ARG;   // 32-bit mask pair index
  if (xhat_sum && xhat_eval_stride != (long long)d_model * ld) return CSN_E_STRIDE;     // the row-sum pass walks dense maps
  if (!dim_ok(d_model)) return CSN_E_DIM;
  if ((ld & 3) || (d_inner & 3) || (n_points & 3)) return CSN_E_ALIGN;
  if (mis16(ctx) || mis16(wfc) || mis16(xres) || mis16(xhat)) return CSN_E_PTR;
  if ((ctx_eval_stride & 3) || (xres_shape_stride & 3) || (xhat_eval_stride & 3)) return CSN_E_STRIDE;
  CsnOutProjArgs a;
  a.ctx = ctx; a.ctx_eval_stride = ctx_eval_stride; a.wfc = wfc;
  a.xres = xres; a.xres_shape_stride = xres_shape_stride; a.res_index = res_index;
  a.xhat = xhat; a.xhat_eval_stride = xhat_eval_stride; a.rstd = rstd;
  a.E = n_evals; a.C = d_model; a.D = d_inner; a.ld = ld; a.n_points = n_points; a.eps = eps;
  a.dropout_p = dropout_p; a.seed = seed;
  a.xhat_sum = xhat_sum; a.sum_ws = sum_ws; a.sum_ws_floats = sum_ws ? sum_ws_floats : 0;
  if (act16() && !act16_fwd_ok()) return CSN_E_ARG;
  a.act16 = act16();
  return csn_launch_outproj_ln_fwd_f32(a, mode(), (hipStream_t)stream);
}

int csn_outproj_ln_bwd_f32(const float* dxhat, const float* xhat, const float* rstd, long long eval_stride,
                           const float* ctx, long long ctx_eval_stride, const float* wfc_t, float* dz, float* dz_res,
                           float* dctx, float* dwfc, float* ws, long long ws_floats, int n_evals, int d_model,
                           int d_inner, int ld, int n_points, int accumulate, float dropout_p,
                           unsigned long long seed, int dctx_split, long long dctx_plane_stride,
                           const float* dxhat_rows, int n_dense_evals, const float* dxhat_scale, int dxhat_group,
                           void* stream) {
  if (dropout_p < 0.f || dropout_p >= 1.f) return CSN_E_ARG;
  if (dxhat_group < 0) return CSN_E_ARG;
  if (dropout_p > 0.f && (long long)(d_model / 2 + 1) * ld >= (1ll << 32)) return CSN_E_ARG;   // 32-bit mask pair index
  if (dctx_split && mode() == 0) return CSN_E_ARG;
  if (n_dense_evals < 0 || n_dense_evals > n_evals || (n_dense_evals > 0 && !dxhat)) return CSN_E_ARG;
  if (!xhat || !rstd || !ctx || !wfc_t || !dz || !dctx || !dwfc || !ws) return CSN_E_ARG;
  if (n_evals <= 0 || n_points <= 0 || d_inner <= 0 || d_model <= 0 || n_points > ld) return CSN_E_ARG;
  if ((ld & 3) || (d_inner & 3) || (d_model & 3) || (n_points & 3)) return CSN_E_ALIGN;
  if (mis16(dxhat) || mis16(xhat) || mis16(ctx) || mis16(wfc_t) || mis16(dz) || mis16(dctx) || mis16(dwfc) || mis16(ws))
    return CSN_E_PTR;
  if ((eval_stride & 3) || (ctx_eval_stride & 3)) return CSN_E_STRIDE;
  hipStream_t st = (hipStream_t)stream;
  CsnLnBwdArgs l;
  l.dxhat = dxhat; l.xhat = xhat; l.rstd = rstd; l.dz = dz; l.dz_res = dz_res; l.eval_stride = eval_stride;
  l.E = n_evals; l.C = d_model; l.ld = ld; l.n_points = n_points;
  l.dropout_p = dropout_p; l.seed = seed;
  l.dxhat_rows = dxhat_rows; l.n_dense = n_dense_evals;
  l.dxhat_scale = dxhat_scale; l.dxhat_group = dxhat_group > 0 ? dxhat_group : 1;
  const int a16 = act16();
  if (a16 && (!act16_bwd_ok() || dctx_split)) return CSN_E_ARG;
  l.act16 = a16;
  int rc = 0;
  // dctx[e][D][n] = wfc_t[D][c] dz[e][c][n]
  if (mode() == 1 && !a16 && !dctx_split && csn_wx_takes(d_inner, d_model) &&
      csn_wx_geometry_takes(csn_dev_lnb_group > 0 && csn_dev_lnb_group < n_evals ? csn_dev_lnb_group : n_evals, n_points, d_inner / 256)) {
    // (development switch CSN_DEV_LNB_GROUP = G > 0: LayerNorm backward and dCtx alternate over groups of G evaluations, so that a
    //  group's dz is read back while it may still sit in the 256 MB Infinity Cache)
    const int G = csn_dev_lnb_group > 0 ? csn_dev_lnb_group : n_evals;
    const bool fused = csn_wx_lnb_takes(l, d_inner) && eval_stride == ctx_eval_stride;
    for (int e0 = 0; e0 < n_evals && !rc; e0 += G) {
      const int ng = n_evals - e0 < G ? n_evals - e0 : G;
      if (fused) {                                       // one pass: xhat in, dz (and dz_res) and dCtx out (wx_lnb.hip)
        CsnWxLnbArgs f{};
        f.w = wfc_t; f.xhat = xhat; f.rstd = rstd; f.eval_stride = eval_stride; f.ld = ld;
        f.dxhat = dxhat; f.dxhat_group = l.dxhat_group; f.n_dense = n_dense_evals; f.dxhat_scale = dxhat_scale; f.dxhat_rows = dxhat_rows;
        f.dz = dz; f.dz_res = dz_res; f.dctx = dctx; f.dctx_eval_stride = ctx_eval_stride;
        f.n_items = ng; f.n_points = n_points; f.e_base = e0; f.dropout_p = dropout_p; f.seed = seed;
        rc = csn_launch_wx_lnb(f, st);
        if (rc != CSN_NOT_TAKEN) continue;
      }
      l.e_base = e0; l.E = ng;
      rc = csn_launch_ln_bwd_f32(l, st);
      if (rc) return rc;
      rc = launch_wx(wfc_t, dz + (long long)e0 * eval_stride, eval_stride, ld, dctx + (long long)e0 * ctx_eval_stride, ctx_eval_stride, ld,
                     d_inner, ng, n_points, 0, 1.f, 0, 0, st);
    }
    if (rc) return rc;
    return wgrad(dz, eval_stride, ld, ctx, ctx_eval_stride, ld, dwfc, d_model, d_inner, n_evals, n_points, 1.f, accumulate, ws,
                 ws_floats, st, 0, 0);
  }
  rc = csn_launch_ln_bwd_f32(l, st);
  if (rc) return rc;
  CsnGemmArgs g;
  g.A = operand(wfc_t, 0, 0, 0, nullptr, d_model);
  g.B = operand(dz, 0, 0, eval_stride, nullptr, ld);
  g.C = operand(dctx, 0, 0, dctx_split ? 2 * ctx_eval_stride : ctx_eval_stride, nullptr, ld);   // split: [eval][2][D][ld]
  g.C.planes = dctx_split; g.C.plane_stride = dctx_plane_stride;
  if (a16) { g.B.fmt = CSN_FMT_16; g.C.planes = 1; }                 // dz in, dctx out: bf16 maps
  g.M = d_inner; g.N = n_points; g.K = d_model;
  g.n0 = 1; g.n1 = 1; g.k_chunk = 0;
  g.alpha = 1.f; g.div_rows = 0; g.div_val = 1.f; g.accumulate = 0; g.eval_ids = nullptr;
  rc = launch_gemm(g, 0, n_evals, st);
  if (rc) return rc;
  // dwfc[c][D] (+)= sum_{e,n} dz[e][c][n] ctx[e][D][n]
  return wgrad(dz, eval_stride, ld, ctx, ctx_eval_stride, ld, dwfc, d_model, d_inner, n_evals, n_points, 1.f,
               accumulate, ws, ws_floats, st, a16 ? CSN_FMT_16 : 0, a16 == 2 ? CSN_FMT_F16_TO_BF16 : (a16 ? CSN_FMT_16 : 0));
}

int csn_project_wgrad_f32(const float* dout, long long dout_shape_stride, int ld_dout, const float* x,
                          long long x_shape_stride, int ld_x, float* dw, int rows, int channels, int n_shapes,
                          int n_points, float scale, int accumulate, float* ws, long long ws_floats,
                          void* stream) {
  if (!dout || !x || !dw || !ws || rows <= 0 || channels <= 0 || n_shapes <= 0 || n_points <= 0) return CSN_E_ARG;
  if (n_points > ld_dout || n_points > ld_x) return CSN_E_ARG;
  if ((ld_dout & 3) || (ld_x & 3) || (n_points & 3)) return CSN_E_ALIGN;
  if (mis16(dout) || mis16(x) || mis16(dw) || mis16(ws)) return CSN_E_PTR;
  if ((dout_shape_stride & 3) || (x_shape_stride & 3)) return CSN_E_STRIDE;
  const bool g16 = act16() && grad16();                               // dout: bf16 gradient maps
  if (g16 && !act16_bwd_ok()) return CSN_E_ARG;
  return wgrad(dout, dout_shape_stride, ld_dout, x, x_shape_stride, ld_x, dw, rows, channels, n_shapes, n_points, scale,
               accumulate, ws, ws_floats, (hipStream_t)stream, g16 ? CSN_FMT_16 : 0, 0);
}

int csn_retrieval_measure_f32(const float* f1, const float* f2, float* out, int s1, int n1, int s2, int n2,
                              int channels, float* ws, long long ws_floats, void* stream) {
  if (!f1 || !f2 || !out || !ws || s1 <= 0 || s2 <= 0 || n1 <= 0 || n2 <= 0 || channels <= 0) return CSN_E_ARG;
  if (channels & 3) return CSN_E_ALIGN;
  if (mis16(f1) || mis16(f2) || mis16(ws)) return CSN_E_PTR;
  const long long need = (long long)s1 * n1 + (long long)s2 * n2 + (long long)s1 * s2 * n1;
  if (ws_floats < need) return CSN_E_WORKSPACE;
  return csn_launch_retrieval_f32(f1, f2, out, s1, n1, s2, n2, channels, ws, (hipStream_t)stream);
}

int csn_rowsum_f32(const float* x, float* out, long long rows, int n_points, long long ld, void* stream) {
  if (!x || !out || rows <= 0 || n_points <= 0 || n_points > ld) return CSN_E_ARG;
  if ((n_points & 3) || (ld & 3)) return CSN_E_ALIGN;
  if (mis16(x)) return CSN_E_PTR;
  return csn_launch_rowsum_f32(x, out, rows, n_points, ld, (hipStream_t)stream, act16());
}

int csn_mix_fwd_f32(const float* xhat, const float* comp, const float* gamma, const float* beta, float* feats,
                    int n_shapes, int k1, int channels, int n_points, const float* xhat_self, void* stream) {
  if (!comp || !gamma || !beta || !feats || n_shapes <= 0 || k1 <= 0 || k1 > 8 || channels <= 0 || n_points <= 0)
    return CSN_E_ARG;
  if (!xhat && !(xhat_self && k1 == 1)) return CSN_E_ARG;
  if (n_points & 3) return CSN_E_ALIGN;
  if (mis16(xhat) || mis16(feats) || mis16(xhat_self)) return CSN_E_PTR;
  return csn_launch_mix_fwd_f32(xhat, comp, gamma, beta, feats, n_shapes, k1, channels, n_points, xhat_self,
                                (hipStream_t)stream, act16());
}

int csn_mix_bwd_f32(const float* dfeats, const float* xhat, const float* comp, const float* gamma, float* dxhat,
                    float* rowdot, float* rowsum, int n_shapes, int k1, int channels, int n_points, const float* xhat_self,
                    float* dxhat_self, void* stream) {
  if (!dfeats || !comp || !gamma || !rowdot || !rowsum) return CSN_E_ARG;
  if (!xhat && !(xhat_self && k1 == 1)) return CSN_E_ARG;
  // gradient maps: all of them or none (none = reductions only; csn_outproj_ln_bwd_f32 then rebuilds them from dfeats)
  const bool want_maps = xhat_self ? dxhat_self != nullptr : dxhat != nullptr;
  if (want_maps && xhat_self && xhat && !dxhat) return CSN_E_ARG;
  if (!want_maps && (dxhat || dxhat_self)) return CSN_E_ARG;
  if (dxhat_self && !xhat_self) return CSN_E_ARG;                     // a gradient map for own-shape maps that were not given
  if (n_shapes <= 0 || k1 <= 0 || k1 > 8 || channels <= 0 || n_points <= 0) return CSN_E_ARG;
  if (n_points & 3) return CSN_E_ALIGN;
  if (mis16(dfeats) || mis16(xhat) || mis16(dxhat) || mis16(xhat_self) || mis16(dxhat_self)) return CSN_E_PTR;
  if (act16() && want_maps) return CSN_E_ARG;                         // fp16 maps: the linked form (reductions only)
  return csn_launch_mix_bwd_f32(dfeats, xhat, comp, gamma, dxhat, rowdot, rowsum, n_shapes, k1, channels, n_points, xhat_self,
                                dxhat_self, (hipStream_t)stream, act16());
}

int csn_compat_fwd_f32(const float* pooled, const float* wq_t, const float* bq, const float* wk_t, const float* bk, float* comp,
                       double* save_u, double* save_norm, int n_shapes, int k1, int channels, int reference_layout, void* stream) {
  if (!pooled || !wq_t || !bq || !wk_t || !bk || !comp || !save_u || !save_norm) return CSN_E_ARG;
  if (n_shapes <= 0 || k1 <= 0 || k1 > 8 || channels <= 0 || channels > 256) return CSN_E_ARG;
  return csn_launch_compat_fwd(pooled, wq_t, bq, wk_t, bk, comp, save_u, save_norm, n_shapes, k1, channels, reference_layout,
                               (hipStream_t)stream);
}

int csn_compat_bwd_f32(const float* dcomp, const float* comp, const double* save_u, const double* save_norm, const float* pooled,
                       const float* wq, const float* wk, double* ws, long long ws_doubles, float* dpooled, float* dwq, float* dbq,
                       float* dwk, float* dbk, int n_shapes, int k1, int channels, int reference_layout, void* stream) {
  if (!dcomp || !comp || !save_u || !save_norm || !pooled || !wq || !wk || !ws || !dpooled || !dwq || !dbq || !dwk || !dbk)
    return CSN_E_ARG;
  if (n_shapes <= 0 || k1 <= 0 || k1 > 8 || channels <= 0 || channels > 256) return CSN_E_ARG;
  const long long rows = (long long)n_shapes * (k1 + 1) * channels;
  if (ws_doubles < 2 * rows) return CSN_E_WORKSPACE;
  return csn_launch_compat_bwd(dcomp, comp, save_u, save_norm, pooled, wq, wk, ws, ws + rows, dpooled, dwq, dbq, dwk, dbk, n_shapes,
                               k1, channels, reference_layout, (hipStream_t)stream);
}

}  // extern "C"
