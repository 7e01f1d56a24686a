// Output projection + residual + LayerNorm of the CSA multi-head attention, and its backward helpers.
//
// Reference arithmetic (MID-FC/csa_models.py:114-118):   q = LayerNorm(fc(ctx) + residual), eps = 1e-6,
// where the residual is the *un-projected* query input (csa_models.py:99).  The affine part of the
// LayerNorm (gamma, beta) is applied by the caller; this kernel emits the normalised activations
// xhat and the reciprocal standard deviation, which is exactly what the backward pass needs.
//
// Orientation: Z^T[c][n] = sum_D W_fc[c][D] * Ctx^T[D][n].  The point index n sits on the lanes and
// ALL channels of a point sit in the registers of one lane pair, so the LayerNorm statistics are an
// in-register reduction (+ one exchange between the two 32-lane halves), the residual x[c][n] is
// read in its native channels-first layout with 128-byte row segments, and xhat is written the same way.
#include "csn_common.h"
#include "csn_kernels.h"

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;
constexpr int BN = 128;          // points per work-group (32 per wave)

template <int CT>
__global__ __launch_bounds__(256, 2) void csn_outproj_ln_fwd_kernel(CsnOutProjArgs p) {
  constexpr int C = 32 * CT;
  __shared__ __attribute__((aligned(16))) float As[C * LDK];     // W_fc[c][k0..k0+31]
  __shared__ __attribute__((aligned(16))) float Bs[BK * BN];     // Ctx^T[k0..k0+31][n0..n0+127]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int e = blockIdx.y, n0 = blockIdx.x * BN;
  const int D = p.D, ld = p.ld, NP = p.n_points;
  const long long rs = p.res_index ? p.res_index[e] : e;

  const csn_rsrc_t Wr = csn_make_rsrc(p.wfc, (long long)C * D * 4);
  const csn_rsrc_t Xr = csn_make_rsrc(p.ctx + (long long)e * p.ctx_eval_stride + n0, ((long long)(D - 1) * ld + (NP - n0)) * 4);
  const csn_rsrc_t Rr = csn_make_rsrc(p.xres + rs * p.xres_shape_stride + n0, ((long long)(C - 1) * ld + (NP - n0)) * 4);
  const csn_rsrc_t Hr = csn_make_rsrc(p.xhat + (long long)e * p.xhat_eval_stride + n0, ((long long)(C - 1) * ld + (NP - n0)) * 4);

  f32x16 acc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  const int pr = tid >> 3, pc = (tid & 7) * 4;        // W slab piece: row pr + 32 i, k piece pc
  const int kr = tid >> 5, kc = (tid & 31) * 4;       // Ctx slab piece: k row kr + 8 i, points kc..kc+3
  unsigned a_off[CT], b_off[4];
#pragma unroll
  for (int i = 0; i < CT; ++i) a_off[i] = (unsigned)((pr + 32 * i) * D + pc) * 4u;
#pragma unroll
  for (int i = 0; i < 4; ++i) b_off[i] = (n0 + kc) < NP ? (unsigned)((kr + 8 * i) * ld + kc) * 4u : CSN_OOB;

  f32x4 ra[CT], rb[4];
  auto load_slab = [&](int k0) {
    const unsigned kp = (k0 + pc) < D ? 0u : CSN_OOB;
#pragma unroll
    for (int i = 0; i < CT; ++i) ra[i] = csn_bload4(Wr, a_off[i] | kp, (unsigned)k0 * 4u);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned kq = (k0 + kr + 8 * i) < D ? 0u : CSN_OOB;
      rb[i] = csn_bload4(Xr, b_off[i] | kq, (unsigned)k0 * (unsigned)ld * 4u);
    }
  };
  auto store_slab = [&]() {
#pragma unroll
    for (int i = 0; i < CT; ++i) *reinterpret_cast<f32x4*>(&As[(pr + 32 * i) * LDK + pc]) = ra[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&Bs[(kr + 8 * i) * BN + kc]) = rb[i];
  };

  const int nk = (D + BK - 1) / BK;
  load_slab(0);
  store_slab();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_slab((kt + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 8) {
      float bf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) bf[t] = Bs[(kk + 4 * h + t) * BN + 32 * wave + l31];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(&As[(c * 32 + l31) * LDK + kk + 4 * h]);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[c] = csn_mfma(af[t], bf[t], acc[c]);
      }
    }
    __syncthreads();
    if (kt + 1 < nk) { store_slab(); __syncthreads(); }
  }

  // ---- + residual, LayerNorm over the C channels of each point (csa_models.py:116-118) -----------
  const int nl = 32 * wave + l31;
  const bool n_ok = (n0 + nl) < NP;
  const unsigned n_off = n_ok ? (unsigned)(4 * h * ld + nl) * 4u : CSN_OOB;
  // dropout on the fc output, before the residual add (csa_models.py:115-116); element index = position in xhat
  if (p.dropout_p > 0.f) {
    const unsigned thr16 = csn_drop_threshold16(p.dropout_p);
    const unsigned salt = csn_block_salt((unsigned long long)e, p.seed);
    const float keep_scale = 1.f / (1.f - p.dropout_p);
    unsigned hp = 0;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (!(r & 1)) hp = csn_fc_pair(c * 32 + csn_acc_row(r, h), (unsigned)ld, (unsigned)(n0 + nl), salt);
        acc[c][r] = csn_keep16(hp, r & 1, thr16) ? acc[c][r] * keep_scale : 0.f;
      }
  }
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[c][r] += csn_bload(Rr, n_off, (unsigned)(c * 32 + csn_acc_row(r, 0)) * (unsigned)ld * 4u);
      sum += acc[c][r];
    }
  sum += csn_xhalf(sum);
  const float mean = sum * (1.f / C);
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float dlt = acc[c][r] - mean;
      sq += dlt * dlt;
    }
  sq += csn_xhalf(sq);
  const float rstd = 1.f / sqrtf(sq * (1.f / C) + p.eps);
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      csn_bstore((acc[c][r] - mean) * rstd, Hr, n_off, (unsigned)(c * 32 + csn_acc_row(r, 0)) * (unsigned)ld * 4u);
  if (n_ok && h == 0) p.rstd[(long long)e * NP + n0 + nl] = rstd;
}

// ---- bf16x3 variant (math mode 1): the contraction runs as three bf16 matrix-core products per fp32 product; the
// residual add, dropout and LayerNorm epilogue are the fp32 code of the exact kernel, unchanged.  Operand split and
// fragment maps: see gemm_bf16x3.hip.
using namespace csn_mode;
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;

// PR = csn_mode::Bf16x3 / Bf16 / F16 (math modes 1 / 2 / 3): three products of hi / lo planes, or one product of one plane
// A16 (single-product modes): Ctx^T arrives as a 16-bit map of the mode's type (staged by copy), xhat leaves as fp16
template <typename PR, int CT, bool A16 = false>
__global__ __launch_bounds__(256, 2) void csn_outproj_ln_fwd_bf16x3_kernel(CsnOutProjArgs p) {
  static_assert(!A16 || PR::NPL == 1, "16-bit activation maps: single-product modes");
  constexpr int NPL = PR::NPL;
  constexpr int XES = A16 ? 2 : 4;                                // bytes per element of Ctx^T and xhat
  constexpr int C = 32 * CT;
  constexpr int PK = BK + 8;                                      // k-contiguous planes: 80-byte rows (conflict-free b128)
  constexpr int PN = BN + 32;                                     // k-major planes: rows 64 B apart mod 256 (conflict-free tr reads)
  __shared__ __attribute__((aligned(16))) short As[NPL][C * PK];   // [plane][c][k]   W_fc[c][k0..k0+31] split into hi / lo
  __shared__ __attribute__((aligned(16))) short Bs[NPL][BK * PN];  // [plane][k][n]   Ctx^T[k0..k0+31][n0..n0+127]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int e = blockIdx.y, n0 = blockIdx.x * BN;
  const int D = p.D, ld = p.ld, NP = p.n_points;
  const long long rs = p.res_index ? p.res_index[e] : e;

  const csn_rsrc_t Wr = csn_make_rsrc(p.wfc, (long long)C * D * 4);
  const csn_rsrc_t Xr = csn_make_rsrc(reinterpret_cast<const char*>(p.ctx) + ((long long)e * p.ctx_eval_stride + n0) * XES,
                                      ((long long)(D - 1) * ld + (NP - n0)) * XES);
  const csn_rsrc_t Rr = csn_make_rsrc(p.xres + rs * p.xres_shape_stride + n0, ((long long)(C - 1) * ld + (NP - n0)) * 4);
  const csn_rsrc_t Hr = csn_make_rsrc(reinterpret_cast<char*>(p.xhat) + ((long long)e * p.xhat_eval_stride + n0) * XES,
                                      ((long long)(C - 1) * ld + (NP - n0)) * XES);

  f32x16 acc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  const int pr = tid >> 3, pc = (tid & 7) * 4;        // W slab piece: row pr + 32 i, k piece pc
  const int kr = tid >> 5, kc = (tid & 31) * 4;       // Ctx slab piece: k row kr + 8 i, points kc..kc+3
  const int kr2 = tid >> 4, ku = tid & 15;            // 16-bit Ctx map: k row kr2 + 16 i, points 8 ku .. 8 ku + 7
  unsigned a_off[CT], b_off[4];
#pragma unroll
  for (int i = 0; i < CT; ++i) a_off[i] = (unsigned)((pr + 32 * i) * D + pc) * 4u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (A16) b_off[i] = (i < 2 && (n0 + 8 * ku) < NP) ? (unsigned)((kr2 + 16 * i) * ld + 8 * ku) * 2u : CSN_OOB;
    else b_off[i] = (n0 + kc) < NP ? (unsigned)((kr + 8 * i) * ld + kc) * 4u : CSN_OOB;
  }

  f32x4 ra[CT], rb[4];
  auto load_slab = [&](int k0) {
    const unsigned kp = (k0 + pc) < D ? 0u : CSN_OOB;
#pragma unroll
    for (int i = 0; i < CT; ++i) ra[i] = csn_bload4(Wr, a_off[i] | kp, (unsigned)k0 * 4u);
    if (A16) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        rb[i] = csn_bload4(Xr, b_off[i] | ((k0 + kr2 + 16 * i) < D ? 0u : CSN_OOB), (unsigned)k0 * (unsigned)ld * 2u);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned kq = (k0 + kr + 8 * i) < D ? 0u : CSN_OOB;
        rb[i] = csn_bload4(Xr, b_off[i] | kq, (unsigned)k0 * (unsigned)ld * 4u);
      }
    }
  };
  auto store_slab = [&]() {
    s16x4 hi, lo;
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      split4<PR>(ra[i], hi, lo);
      *reinterpret_cast<s16x4*>(&As[0][(pr + 32 * i) * PK + pc]) = hi;
      if constexpr (NPL == 2) *reinterpret_cast<s16x4*>(&As[NPL - 1][(pr + 32 * i) * PK + pc]) = lo;
    }
    if (A16) {
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(&Bs[0][(kr2 + 16 * i) * PN + 8 * ku]) = rb[i];
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      split4<PR>(rb[i], hi, lo);
      *reinterpret_cast<s16x4*>(&Bs[0][(kr + 8 * i) * PN + kc]) = hi;
      if constexpr (NPL == 2) *reinterpret_cast<s16x4*>(&Bs[NPL - 1][(kr + 8 * i) * PN + kc]) = lo;
    }
  };
  // transposing read of the k-major Ctx planes: lane group g = lane >> 4 covers points 16 (g & 1) .. +15 of this wave's 32
  // and k rows 8 (g >> 1) .. +7; inside the group lane 4 q + p addresses row q, points 4 p .. 4 p + 3
  const int tr_base = (8 * (lane >> 5) + ((lane >> 2) & 3)) * PN + 32 * wave + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  const int nk = (D + BK - 1) / BK;
  load_slab(0);
  store_slab();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_slab((kt + 1) * BK);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int o = tr_base + 16 * s * PN;
      const s16x8 bh = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(&Bs[0][o])),
                             __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(&Bs[0][o + 4 * PN])));
      s16x8 bl = bh;
      if constexpr (NPL == 2)
        bl = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(&Bs[1][o])),
                   __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(&Bs[1][o + 4 * PN])));
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int a = (c * 32 + l31) * PK + 16 * s + 8 * h;
        const s16x8 ah = *reinterpret_cast<const s16x8*>(&As[0][a]);
        const s16x8 al = *reinterpret_cast<const s16x8*>(&As[NPL - 1][a]);
        if constexpr (PR::NT == 3) {
          acc[c] = mfma32<PR::HALF>(al, bh, acc[c]);                                     // small terms first
          acc[c] = mfma32<PR::HALF>(ah, bl, acc[c]);
        }
        acc[c] = mfma32<PR::HALF>(ah, bh, acc[c]);
      }
    }
    __syncthreads();
    if (kt + 1 < nk) { store_slab(); __syncthreads(); }
  }

  // ---- + residual, LayerNorm over the C channels of each point (csa_models.py:116-118) -----------
  const int nl = 32 * wave + l31;
  const bool n_ok = (n0 + nl) < NP;
  const unsigned n_off = n_ok ? (unsigned)(4 * h * ld + nl) * 4u : CSN_OOB;
  // dropout on the fc output, before the residual add (csa_models.py:115-116); element index = position in xhat
  if (p.dropout_p > 0.f) {
    const unsigned thr16 = csn_drop_threshold16(p.dropout_p);
    const unsigned salt = csn_block_salt((unsigned long long)e, p.seed);
    const float keep_scale = 1.f / (1.f - p.dropout_p);
    unsigned hp = 0;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (!(r & 1)) hp = csn_fc_pair(c * 32 + csn_acc_row(r, h), (unsigned)ld, (unsigned)(n0 + nl), salt);
        acc[c][r] = csn_keep16(hp, r & 1, thr16) ? acc[c][r] * keep_scale : 0.f;
      }
  }
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[c][r] += csn_bload(Rr, n_off, (unsigned)(c * 32 + csn_acc_row(r, 0)) * (unsigned)ld * 4u);
      sum += acc[c][r];
    }
  sum += csn_xhalf(sum);
  const float mean = sum * (1.f / C);
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float dlt = acc[c][r] - mean;
      sq += dlt * dlt;
    }
  sq += csn_xhalf(sq);
  const float rstd = 1.f / sqrtf(sq * (1.f / C) + p.eps);
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float xh = (acc[c][r] - mean) * rstd;
      if constexpr (A16) csn_bstore16(to16<true>(xh), Hr, n_ok ? n_off >> 1 : CSN_OOB, (unsigned)(c * 32 + csn_acc_row(r, 0)) * (unsigned)ld * 2u);
      else csn_bstore(xh, Hr, n_off, (unsigned)(c * 32 + csn_acc_row(r, 0)) * (unsigned)ld * 4u);
    }
  if (n_ok && h == 0) p.rstd[(long long)e * NP + n0 + nl] = rstd;
}

// LayerNorm backward without the affine part:  dz = rstd * (dx - mean_c(dx) - xhat * mean_c(dx * xhat)).
// Single pass over HBM: a work-group owns 64 points (lanes along n: 256-byte row segments); its 4 waves split the
// channels (wave g takes c = g C/4 + i), every thread keeps its C/4 (dx, xhat) pairs in registers, the two channel sums
// are combined across the 4 waves through LDS, and dz is produced from the registers.
// dx = dxhat (evaluations below n_dense only) + dxhat_rows[e][c] (constant along the points).  All loads are branch-free
// (buffer descriptors; an absent dense part is a zero-sized window), so the 2 C/4 loads of a thread are in flight together.
// A16 (16-bit activation maps): xhat is read as an fp16 map, dz leaves as a bf16 map; dxhat and dz_res stay fp32.
template <int CPT, bool A16 = false>
__global__ __launch_bounds__(256) void csn_ln_bwd_kernel(CsnLnBwdArgs p) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = blockIdx.x * 64 + lane;
  const int e = blockIdx.y + p.e_base;
  const bool ok = n < p.n_points;
  constexpr int C = 4 * CPT;
  const long long win = ((long long)(C - 1) * p.ld + p.n_points) * 4;
  const bool dense = e < p.n_dense;
  const int src = p.dxhat_group > 1 ? e / p.dxhat_group : e;
  const csn_rsrc_t Gr = csn_make_rsrc(dense ? p.dxhat + (long long)src * p.eval_stride : nullptr, dense ? win : 0);
  const csn_rsrc_t Xr = A16 ? csn_make_rsrc(reinterpret_cast<const short*>(p.xhat) + (long long)e * p.eval_stride, win / 2)
                            : csn_make_rsrc(p.xhat + (long long)e * p.eval_stride, win);
  const unsigned voff = ok ? (unsigned)n * 4u : CSN_OOB;
  const unsigned voff16 = ok ? (unsigned)n * 2u : CSN_OOB;
  const unsigned ldb = (unsigned)p.ld * 4u;
  float gx[CPT], xx[CPT], rw[CPT], sc[CPT];
  if (p.dxhat_scale && dense) {
    const float* __restrict__ scl = p.dxhat_scale + (long long)e * C + g * CPT;       // wave-uniform: scalar loads
#pragma unroll
    for (int i = 0; i < CPT; ++i) sc[i] = scl[i];
  } else {
#pragma unroll
    for (int i = 0; i < CPT; ++i) sc[i] = 1.f;
  }
  if (p.dxhat_rows) {
    const float* __restrict__ rows = p.dxhat_rows + (long long)e * C + g * CPT;      // wave-uniform: scalar loads
#pragma unroll
    for (int i = 0; i < CPT; ++i) rw[i] = rows[i];
  } else {
#pragma unroll
    for (int i = 0; i < CPT; ++i) rw[i] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    gx[i] = csn_bload(Gr, voff, (unsigned)(g * CPT + i) * ldb);
    if constexpr (A16)
      xx[i] = csn_mode::from16<true>((short)__builtin_amdgcn_raw_buffer_load_b16(Xr, voff16, (unsigned)(g * CPT + i) * (ldb >> 1), 0));
    else xx[i] = csn_bload(Xr, voff, (unsigned)(g * CPT + i) * ldb);
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    gx[i] = fmaf(gx[i], sc[i], rw[i]);
    s1 += gx[i];
    s2 += gx[i] * xx[i];
  }
  red[0][g][lane] = s1;
  red[1][g][lane] = s2;
  __syncthreads();
  const float m1 = (red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]) / C;
  const float m2 = (red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]) / C;
  const float rstd = ok ? p.rstd[(long long)e * p.n_points + n] : 0.f;
  const bool drop = p.dropout_p > 0.f;
  const unsigned thr16 = csn_drop_threshold16(p.dropout_p);
  const unsigned salt = drop ? csn_block_salt((unsigned long long)e, p.seed) : 0u;
  const float keep_scale = drop ? 1.f / (1.f - p.dropout_p) : 1.f;
  const csn_rsrc_t Zr = A16 ? csn_make_rsrc(reinterpret_cast<short*>(p.dz) + (long long)e * p.eval_stride, win / 2)
                            : csn_make_rsrc(p.dz + (long long)e * p.eval_stride, win);
  const csn_rsrc_t Zres = csn_make_rsrc(p.dz_res ? p.dz_res + (long long)e * p.eval_stride : nullptr, p.dz_res ? win : 0);
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    gx[i] = rstd * (gx[i] - m1 - xx[i] * m2);
    csn_bstore(gx[i], Zres, voff, (unsigned)(g * CPT + i) * ldb);     // the residual branch sees no mask (vanishes when absent)
  }
  if (drop) {                                                         // one uniform branch, not one per element
    unsigned hp = 0;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {                                   // CPT even: channels g CPT + i, + 1 (i even) are one pair
      if ((CPT & 1) || !(i & 1)) hp = csn_fc_pair(g * CPT + i, (unsigned)p.ld, (unsigned)n, salt);
      gx[i] = csn_keep16(hp, (g * CPT + i) & 1, thr16) ? gx[i] * keep_scale : 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    if constexpr (A16) csn_bstore16(csn_mode::to16<false>(gx[i]), Zr, voff16, (unsigned)(g * CPT + i) * (ldb >> 1));
    else csn_bstore(gx[i], Zr, voff, (unsigned)(g * CPT + i) * ldb);
  }
}

// out[e][h][n] = sum_{c < d} a[e][h*d + c][n] * b[e][h*d + c][n]    (delta = rowsum(dO * O) of the softmax backward)
// a may be a split tensor (bf16 hi plane at a, lo plane a_plane_stride bf16 elements later; same strides in elements)
__global__ __launch_bounds__(256) void csn_rowdot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ out, const int* __restrict__ eval_ids, int H,
                                                         int d, int ld, int n_points, long long eval_stride, int a_split,
                                                         long long a_plane_stride) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int hd = blockIdx.y % H;
  const int e = eval_ids ? eval_ids[blockIdx.y / H] : (int)(blockIdx.y / H);
  if (n >= n_points) return;
  const long long base = (long long)e * eval_stride + (long long)hd * d * ld + n;
  float s = 0.f;
  if (a_split) {
    // split layout [eval][2 planes][rows][ld]: the evaluation stride doubles
    const __bf16* __restrict__ ah = reinterpret_cast<const __bf16*>(a) + (long long)e * eval_stride;
    const __bf16* __restrict__ al = ah + a_plane_stride;
    for (int c = 0; c < d; ++c) {
      const long long i = base + (long long)c * ld;
      s += ((float)ah[i] + (float)al[i]) * b[i];
    }
  } else {
    for (int c = 0; c < d; ++c) s += a[base + (long long)c * ld] * b[base + (long long)c * ld];
  }
  out[((long long)e * H + hd) * n_points + n] = s;
}

template <int CT>
int launch_fwd(const CsnOutProjArgs& a, int mode, hipStream_t st) {
  dim3 grid((a.n_points + BN - 1) / BN, a.E);
  if (a.act16) {
    if (mode == 2) hipLaunchKernelGGL((csn_outproj_ln_fwd_bf16x3_kernel<Bf16, CT, true>), grid, dim3(256), 0, st, a);
    else if (mode == 3) hipLaunchKernelGGL((csn_outproj_ln_fwd_bf16x3_kernel<F16, CT, true>), grid, dim3(256), 0, st, a);
    else return -1;
    return (int)hipGetLastError();
  }
  if (mode == 1) hipLaunchKernelGGL((csn_outproj_ln_fwd_bf16x3_kernel<Bf16x3, CT>), grid, dim3(256), 0, st, a);
  else if (mode == 2) hipLaunchKernelGGL((csn_outproj_ln_fwd_bf16x3_kernel<Bf16, CT>), grid, dim3(256), 0, st, a);
  else if (mode == 3) hipLaunchKernelGGL((csn_outproj_ln_fwd_bf16x3_kernel<F16, CT>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((csn_outproj_ln_fwd_kernel<CT>), grid, dim3(256), 0, st, a);
  return (int)hipGetLastError();
}

}  // namespace

int csn_launch_outproj_ln_fwd_f32(const CsnOutProjArgs& a, int mode, hipStream_t st) {
  if (a.E <= 0 || a.n_points <= 0) return 0;
  if ((a.ld & 3) || (a.D & 3) || (a.n_points & 3)) return -2;
  if ((a.ctx_eval_stride & 3) || (a.xres_shape_stride & 3) || (a.xhat_eval_stride & 3)) return -4;
  if (mode && a.C == 256 && a.n_points >= 224) return csn_launch_outproj_ln_big(a, mode, st);     // 256 x 256 tiles (gemm_bf16x3.hip)
  int rc;
  switch (a.C) {
    case 32: rc = launch_fwd<1>(a, mode, st); break;
    case 64: rc = launch_fwd<2>(a, mode, st); break;
    case 96: rc = launch_fwd<3>(a, mode, st); break;
    case 128: rc = launch_fwd<4>(a, mode, st); break;
    case 256: rc = launch_fwd<8>(a, mode, st); break;
    default: return -5;
  }
  if (rc || !a.xhat_sum) return rc;
  return csn_launch_rowsum_f32(a.xhat, a.xhat_sum, (long long)a.E * a.C, a.n_points, a.ld, st, a.act16);   // streaming pass over xhat
}

int csn_launch_ln_bwd_f32(const CsnLnBwdArgs& a, hipStream_t st) {
  if (a.E <= 0 || a.n_points <= 0) return 0;
  dim3 grid((a.n_points + 63) / 64, a.E);
  if (a.act16) {
    switch (a.C) {
      case 32: hipLaunchKernelGGL((csn_ln_bwd_kernel<8, true>), grid, dim3(256), 0, st, a); break;
      case 64: hipLaunchKernelGGL((csn_ln_bwd_kernel<16, true>), grid, dim3(256), 0, st, a); break;
      case 96: hipLaunchKernelGGL((csn_ln_bwd_kernel<24, true>), grid, dim3(256), 0, st, a); break;
      case 128: hipLaunchKernelGGL((csn_ln_bwd_kernel<32, true>), grid, dim3(256), 0, st, a); break;
      case 256: hipLaunchKernelGGL((csn_ln_bwd_kernel<64, true>), grid, dim3(256), 0, st, a); break;
      default: return -5;
    }
    return (int)hipGetLastError();
  }
  switch (a.C) {
    case 32: hipLaunchKernelGGL((csn_ln_bwd_kernel<8>), grid, dim3(256), 0, st, a); break;
    case 64: hipLaunchKernelGGL((csn_ln_bwd_kernel<16>), grid, dim3(256), 0, st, a); break;
    case 96: hipLaunchKernelGGL((csn_ln_bwd_kernel<24>), grid, dim3(256), 0, st, a); break;
    case 128: hipLaunchKernelGGL((csn_ln_bwd_kernel<32>), grid, dim3(256), 0, st, a); break;
    case 256: hipLaunchKernelGGL((csn_ln_bwd_kernel<64>), grid, dim3(256), 0, st, a); break;
    default: return -5;
  }
  return (int)hipGetLastError();
}

int csn_launch_rowdot_f32(const float* a, const float* b, float* out, const int* eval_ids, int E, int H, int d, int ld,
                          int n_points, long long eval_stride, int a_split, long long a_plane_stride, hipStream_t st) {
  if (E <= 0 || n_points <= 0) return 0;
  dim3 grid((n_points + 255) / 256, E * H);
  hipLaunchKernelGGL(csn_rowdot_kernel, grid, dim3(256), 0, st, a, b, out, eval_ids, H, d, ld, n_points, eval_stride, a_split,
                     a_plane_stride);
  return (int)hipGetLastError();
}
