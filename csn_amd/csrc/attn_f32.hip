// Fused block-diagonal attention for the CSA layer, exact fp32 on the CDNA4 matrix cores.
//
// Reference arithmetic: ScaledDotProductAttention.forward (MID-FC/csa_models.py:138-144) applied to
// query chunk i against key/value chunk i only (csa_models.py:86-90, T = 500 points per chunk).
//
// Everything is channel-major, so one skeleton serves forward and backward:
//
//   R      [d][q]   per-wave register operand, 32 query points on the lanes   (fwd: Qs^T   bwd: dO^T)
//   tileA  [d][key] 32 keys of the streamed operand, k-major in LDS           (fwd: K^T    bwd: V^T)
//   tileB  [d][key] 32 keys, key-contiguous in LDS (16-byte fragment reads)   (fwd: V^T    bwd: K^T)
//
//   phase 1   T1[key][q] = sum_d tileA[d][key] * R[d][q]        fwd: S^T = K Qs^T      bwd: dP^T = V dO^T
//   pointwise                                                   fwd: online softmax     bwd: dS = P (dP - delta)
//   phase 2   OUT[d][q] += sum_key tileB[d][key] * T1[key][q]   fwd: O^T += V^T P^T     bwd: dQs^T += K^T dS^T
//
// The phase-1 accumulator (key on registers, query on the lane) is *already* the B operand of the
// phase-2 products, so the 32 x 32 score tile never leaves registers, and all softmax statistics
// are per-lane scalars (no cross-lane traffic except one exchange between the two 32-lane halves).
//
// Forward additionally writes the raw scores S^T to HBM when training: at 64 FLOP/clk/SIMD the fp32
// matrix pipe makes recomputing S (2*T*T*d flops per block) dearer than the 4*T*T bytes it costs to
// keep it in the 288 GB of HBM3E.  Backward turns S^T into P^T in place and emits dS^T next to it;
// dK^T / dV^T are then plain batched GEMMs over those two buffers (gemm_f32.hip).
//
// Work-group = 4 waves = 128 query points of one (evaluation, head, block); one wave per SIMD
// (the wave keeps R and OUT, 2 x d/2 registers per lane, resident for the whole key sweep).
#include "csn_common.h"
#include "csn_kernels.h"

namespace {

constexpr int KT = 32;               // keys per streamed tile
constexpr int LDB = KT + 4;          // padded row of tileB (conflict-free ds_read_b128)

template <int DT, bool BWD>
__global__ __launch_bounds__(256, 1) void csn_attn_f32_kernel(CsnAttnArgs p) {
  constexpr int D = 32 * DT;
  // two stages of each streamed tile: stage (kt & 1) is read while stage ((kt + 1) & 1) is filled
  __shared__ __attribute__((aligned(16))) float tileA[2][D * KT];
  __shared__ __attribute__((aligned(16))) float tileB[2][D * LDB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int e = blockIdx.z;
  const int hd = blockIdx.y % p.H, blk = blockIdx.y / p.H;
  const int T = p.T, Tp = p.Tp, ld = p.ld;
  const int qrow = blockIdx.x * 128 + wave * 32 + l31;        // query index inside the block
  const bool q_ok = qrow < T;

  const long long qs = p.q_index ? p.q_index[e] : e;
  const long long ks = p.kv_index ? p.kv_index[e] : e;
  const long long head_off = (long long)hd * D * ld + (long long)blk * T;
  const long long win = ((long long)(D - 1) * ld + T) * 4;     // bytes spanned by a [D][T] window of pitch ld
  const csn_rsrc_t Rr = csn_make_rsrc(p.q + qs * p.q_shape_stride + head_off, win);
  const csn_rsrc_t Ar = csn_make_rsrc((BWD ? p.v : p.k) + ks * p.kv_shape_stride + head_off, win);
  const csn_rsrc_t Br = csn_make_rsrc((BWD ? p.k : p.v) + ks * p.kv_shape_stride + head_off, win);
  const csn_rsrc_t Or = csn_make_rsrc(p.out + (long long)e * p.out_eval_stride + head_off, win);
  const long long stat_off = ((long long)e * p.H + hd) * ((long long)p.n_blocks * T) + (long long)blk * T;
  const long long sc_off = (((long long)e * p.H + hd) * p.n_blocks + blk) * ((long long)T * Tp);
  const bool have_scores = p.scores != nullptr;
  const csn_rsrc_t Sr = csn_make_rsrc(have_scores ? p.scores + sc_off : nullptr, have_scores ? (long long)T * Tp * 4 : 0);
  const csn_rsrc_t dSr = csn_make_rsrc(BWD ? p.dscores + sc_off : nullptr, BWD ? (long long)T * Tp * 4 : 0);

  // per-lane byte offsets (switched off for query rows beyond the block)
  // (scalar offsets handed to the buffer instructions must be wave-uniform, so everything that
  //  depends on the lane half h lives in the per-lane offset)
  const unsigned q_off = q_ok ? (unsigned)qrow * 4u : CSN_OOB;
  const unsigned r_off = q_ok ? (unsigned)(h * ld + qrow) * 4u : CSN_OOB;        // row 2 s + h
  const unsigned o_off = q_ok ? (unsigned)(4 * h * ld + qrow) * 4u : CSN_OOB;    // row .. + 4 h

  // ---- register-resident operand R[d][q]: lane (q, h) keeps rows d = 2 s + h ------------------
  float R[D / 2];
#pragma unroll
  for (int s = 0; s < D / 2; ++s) R[s] = csn_bload(Rr, r_off, (unsigned)(2 * s) * ld * 4u);

  f32x16 O[DT];
#pragma unroll
  for (int c = 0; c < DT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[c][r] = 0.f;

  float m_run = -INFINITY, l_run = 0.f;       // forward: running max / partial sum of this lane half
  float lse_q = 0.f, delta_q = 0.f;           // backward: per-query constants
  if (BWD) {
    lse_q = q_ok ? p.lse[stat_off + qrow] : 0.f;
    delta_q = q_ok ? p.delta[stat_off + qrow] : 0.f;
  }

  // ---- streamed tiles: global -> registers -> LDS ----------------------------------------------
  // thread t moves, for i < DT, the 16 bytes at row (t + 256 i) / 8, keys 4 ((t + 256 i) % 8) ..+3
  unsigned t_off[DT];
  const int t_c = (tid & 7) * 4;
#pragma unroll
  for (int i = 0; i < DT; ++i) t_off[i] = (unsigned)(((tid + 256 * i) >> 3) * ld + t_c) * 4u;

  f32x4 g[DT];
  auto fetch = [&](const csn_rsrc_t& rs, int kt) {
    const int k0 = kt * KT;
    // T % 4 == 0: a 16-byte piece is all in or all out; pieces past the block end are switched off
    const unsigned poison = (k0 + t_c) < T ? 0u : CSN_OOB;
#pragma unroll
    for (int i = 0; i < DT; ++i) g[i] = csn_bload4(rs, t_off[i] | poison, (unsigned)k0 * 4u);
  };
  auto commitA = [&](int st) {
#pragma unroll
    for (int i = 0; i < DT; ++i)
      *reinterpret_cast<f32x4*>(&tileA[st][((tid + 256 * i) >> 3) * KT + t_c]) = g[i];
  };
  auto commitB = [&](int st) {
#pragma unroll
    for (int i = 0; i < DT; ++i)
      *reinterpret_cast<f32x4*>(&tileB[st][((tid + 256 * i) >> 3) * LDB + t_c]) = g[i];
  };

  const int nkt = (T + KT - 1) / KT;
  fetch(Ar, 0); commitA(0);
  fetch(Br, 0); commitB(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1, nxt = cur ^ 1;
    const bool more = kt + 1 < nkt;
    if (more) fetch(Ar, kt + 1);

    // backward: the saved scores of this tile, requested early so they land under phase 1
    float sv[16];
    unsigned s_off[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * KT + csn_acc_row(r, h);
      s_off[r] = (q_ok && key < T) ? (unsigned)(key * Tp + qrow) * 4u : CSN_OOB;
    }
    if (BWD) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = csn_bload(Sr, s_off[r]);
    }

    // ---- phase 1: T1[key][q] = sum_d tileA[d][key] R[d][q] ------------------------------------
    f32x16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
    const float* __restrict__ tA = &tileA[cur][h * KT + l31];
#pragma unroll
    for (int s = 0; s < D / 2; ++s) S = csn_mfma(tA[2 * s * KT], R[s], S);

    if (more) { commitA(nxt); fetch(Br, kt + 1); }

    // ---- pointwise ----------------------------------------------------------------------------
    if (!BWD) {
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * KT + csn_acc_row(r, h);
        if (key >= T) S[r] = -INFINITY;
        csn_bstore(S[r], Sr, s_off[r]);             // (zero-sized window when scores are not kept)
        mx = fmaxf(mx, S[r]);
      }
      mx = fmaxf(mx, csn_xhalf(mx));
      // lazy rescale: only when some query's running maximum would grow by more than the threshold
      if (__any(mx > m_run + p.rescale_threshold)) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = (m_new == -INFINITY) ? 1.f : expf(m_run - m_new);
#pragma unroll
        for (int c = 0; c < DT; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) O[c][r] *= alpha;
        l_run *= alpha;
        m_run = m_new;
      }
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = expf(S[r] - m_run);        // masked keys: exp(-inf) = 0
        S[r] = pv;
        ps += pv;
      }
      l_run += ps;
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = s_off[r] != CSN_OOB;
        const float pv = ok ? expf(sv[r] - lse_q) : 0.f;          // softmax probability (csa_models.py:141)
        const float ds = pv * (S[r] - delta_q);                    // d softmax
        csn_bstore(pv, Sr, s_off[r]);
        csn_bstore(ds, dSr, s_off[r]);
        S[r] = ds;
      }
    }

    // ---- phase 2: OUT[c][q] += sum_key tileB[c][key] T1[key][q] --------------------------------
    const float* __restrict__ tB = &tileB[cur][l31 * LDB + 4 * h];
#pragma unroll
    for (int c = 0; c < DT; ++c) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(&tB[c * 32 * LDB + 8 * gq]);
#pragma unroll
        for (int i = 0; i < 4; ++i) O[c] = csn_mfma(a[i], S[4 * gq + i], O[c]);
      }
    }

    if (more) commitB(nxt);
    __syncthreads();
  }

  // ---- epilogue -----------------------------------------------------------------------------------
  float inv = 1.f;
  if (!BWD) {
    const float l_tot = l_run + csn_xhalf(l_run);
    inv = 1.f / l_tot;
    if (q_ok && h == 0 && p.lse) p.lse[stat_off + qrow] = m_run + logf(l_tot);
  }
#pragma unroll
  for (int c = 0; c < DT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      csn_bstore(O[c][r] * inv, Or, o_off, (unsigned)(c * 32 + csn_acc_row(r, 0)) * ld * 4u);
}

template <int DT>
int launch_dt(const CsnAttnArgs& a, bool bwd, hipStream_t st) {
  dim3 grid((a.T + 127) / 128, a.n_blocks * a.H, a.E);
  if (bwd) hipLaunchKernelGGL((csn_attn_f32_kernel<DT, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((csn_attn_f32_kernel<DT, false>), grid, dim3(256), 0, st, a);
  return (int)hipGetLastError();
}

int launch_any(const CsnAttnArgs& a, int d, bool bwd, hipStream_t st) {
  if (a.E <= 0 || a.n_blocks <= 0) return 0;
  if ((a.T & 3) || (a.ld & 3) || (a.Tp & 3)) return -2;
  if ((a.q_shape_stride & 3) || (a.kv_shape_stride & 3)) return -4;
  switch (d) {
    case 32: return launch_dt<1>(a, bwd, st);
    case 64: return launch_dt<2>(a, bwd, st);
    case 96: return launch_dt<3>(a, bwd, st);
    case 128: return launch_dt<4>(a, bwd, st);
    case 256: return launch_dt<8>(a, bwd, st);
    default: return -5;
  }
}

}  // namespace

int csn_launch_attn_fwd_f32(const CsnAttnArgs& a, int d, hipStream_t st) { return launch_any(a, d, false, st); }
int csn_launch_attn_bwd_f32(const CsnAttnArgs& a, int d, hipStream_t st) { return launch_any(a, d, true, st); }
