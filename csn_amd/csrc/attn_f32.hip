// Fused block-diagonal attention for the CSA layer, exact fp32 on the CDNA4 matrix cores.
//
// Reference arithmetic: ScaledDotProductAttention.forward (MID-FC/csa_models.py:138-144) applied to
// query chunk i against key/value chunk i only (csa_models.py:86-90, T = 500 points per chunk).
//
// Everything is channel-major, so one skeleton serves forward and backward:
//
//   R      [d][q]   per-wave register operand, 16 query points on the lanes   (fwd: Qs^T   bwd: dO^T)
//   tileA  [d][key] 32 keys of the streamed operand, k-major in LDS           (fwd: K^T    bwd: V^T)
//   tileB  [d][key] 32 keys, key-contiguous in LDS (16-byte fragment reads)   (fwd: V^T    bwd: K^T)
//
//   phase 1   T1[key][q] = sum_d tileA[d][key] * R[d][q]        fwd: S^T = K Qs^T      bwd: dP^T = V dO^T
//   pointwise                                                   fwd: online softmax     bwd: dS = P (dP - delta)
//   phase 2   OUT[d][q] += sum_key tileB[d][key] * T1[key][q]   fwd: O^T += V^T P^T     bwd: dQs^T += K^T dS^T
//
// The phase-1 accumulator (key on registers, query on the lane) is *already* the B operand of the
// phase-2 products, so the score tile never leaves registers, and all softmax statistics are per-lane
// scalars (two lane exchanges, l ^ 16 and l ^ 32, combine the four key quarters of a query).
//
// Matrix instruction: v_mfma_f32_16x16x4_f32 (A: lane l = A[l & 15][l >> 4], B: lane l = B[l >> 4][l & 15],
// C: reg r of lane l = C[4 (l >> 4) + r][l & 15]).  A wave owns 16 queries, which halves the register
// footprint of R and OUT (64 + 64 registers at d = 256) against the 32x32x2 shape and lets TWO waves share a
// SIMD: while one wave sits in its softmax / LDS / barrier phases the other keeps the fp32 matrix pipe busy
// (round-1 profile of the 32-query, one-wave-per-SIMD version: pipe 66 % busy, 25 % of wave time in waits).
//
// Forward additionally writes the raw scores S^T to HBM when training: at 64 FLOP/clk/SIMD the fp32
// matrix pipe makes recomputing S (2*T*T*d flops per block) dearer than the 4*T*T bytes it costs to
// keep it in the 288 GB of HBM3E.  Backward turns S^T into P^T in place and emits dS^T next to it;
// dK^T / dV^T are then plain batched GEMMs over those two buffers (gemm_f32.hip).
//
// Work-group = 8 waves = 128 query points of one (evaluation, head, block).  LDS: two stages of both
// tiles, un-padded 128-byte rows made conflict-free by XOR swizzles (tileA: key half ^= row parity;
// tileB: 16-byte chunk ^= (row >> 1) & 7), 128 KB at d = 256.
#include "csn_common.h"
#include "csn_kernels.h"

namespace {

constexpr int KT = 32;               // keys per streamed tile

typedef float f32x4v __attribute__((ext_vector_type(4)));

CSN_DEVINL f32x4v mfma16(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int DT, bool BWD>
__global__ __launch_bounds__(512, 2) void csn_attn_f32_kernel(CsnAttnArgs p) {
  constexpr int D = 32 * DT;
  constexpr int PIECES = D * 8;                         // 16-byte pieces per streamed tile
  constexpr int NP_T = (PIECES + 511) / 512;            // pieces per thread per tile
  // [A | B][stage][row][32 keys] — one array, so that the prologue / epilogue can use all of it as a [D][128 queries] block
  __shared__ __attribute__((aligned(16))) float tiles[2][2][D * KT];
  auto& tileA = tiles[0];
  auto& tileB = tiles[1];
  static_assert(sizeof(tiles) >= D * 128 * 4, "staging block");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 15, kq = lane >> 4;
  // XCD-aware work-group order.  A unit = one (evaluation, head, block); its QT query tiles all stream the same K/V
  // block, so they should share one XCD's L2.  Work-groups are dealt round-robin over the 8 XCDs, hence the QT tiles of
  // unit u get ids 8 * (QT * (u / 8) + qt) + (u % 8): same residue mod 8 (same XCD), adjacent in dispatch order.
  // (Placement only changes speed: every tile is self-contained.)
  const int Tq = p.Tq > 0 ? p.Tq : p.T;                       // queries per block (T: keys per block)
  const int QT = (Tq + 127) / 128;
  const int Y = p.n_blocks * p.H;
  const int L = blockIdx.x, slot = L & 7, jj = L >> 3;
  const int qt = jj % QT, u = (jj / QT) * 8 + slot;
  if (u >= Y * p.E) return;
  const int e = p.eval_ids ? p.eval_ids[u / Y] : u / Y;
  const int hd = (u % Y) % p.H, blk = (u % Y) / p.H;
  // ragged batches: this evaluation's own query / key counts (Tq, p.T stay the maxima that lay out the buffers)
  const bool short_blk = p.T_last > 0 && blk == p.n_blocks - 1;     // the row ends inside the last block
  const int Tq_e = p.tq_arr ? p.tq_arr[e] : (short_blk ? p.T_last : Tq);
  if (qt * 128 >= Tq_e) return;                                // a query tile beyond a short evaluation (whole work-group)
  const int T = p.t_arr ? p.t_arr[e] : (short_blk ? p.T_last : p.T), Tp = p.Tp, ld = p.ld, ldk = p.ld_kv > 0 ? p.ld_kv : p.ld;
  const int qrow = qt * 128 + wave * 16 + lq;                  // query index inside the block
  const bool q_ok = qrow < Tq_e;

  const long long qs = p.q_index ? p.q_index[e] : e;
  const long long ks = p.kv_index ? p.kv_index[e] : e;
  const long long os = p.out_index ? p.out_index[e] : e;
  const long long head_off = (long long)hd * D * ld + (long long)blk * Tq;
  const long long win = ((long long)(D - 1) * ld + Tq) * 4;    // bytes spanned by a [D][Tq] window of pitch ld
  const long long head_off_kv = (long long)hd * D * ldk + (long long)blk * p.T;   // (p.T: the layout; T may be a short last block)
  const long long win_kv = ((long long)(D - 1) * ldk + (T + 3) / 4 * 4) * 4;
  const csn_rsrc_t Rr = csn_make_rsrc(p.q + qs * p.q_shape_stride + head_off, win);
  const csn_rsrc_t Ar = csn_make_rsrc((BWD ? p.v : p.k) + ks * p.kv_shape_stride + head_off_kv, win_kv);
  const csn_rsrc_t Br = csn_make_rsrc((BWD ? p.k : p.v) + ks * p.kv_shape_stride + head_off_kv, win_kv);
  const csn_rsrc_t Or = csn_make_rsrc(p.out + os * p.out_eval_stride + head_off, win);
  const long long stat_off = ((long long)e * p.H + hd) * ((long long)p.n_blocks * Tq) + (long long)blk * Tq;
  const long long sc_off = (((long long)e * p.H + hd) * p.n_blocks + blk) * ((long long)Tq * Tp);
  const bool have_scores = p.scores != nullptr;
  const csn_rsrc_t Sr = csn_make_rsrc(have_scores ? p.scores + sc_off : nullptr, have_scores ? (long long)Tq * Tp * 4 : 0);
  const csn_rsrc_t dSr = csn_make_rsrc(BWD ? p.dscores + sc_off : nullptr, BWD ? (long long)Tq * Tp * 4 : 0);

  // per-lane byte offsets (scalar offsets handed to the buffer instructions must be wave-uniform, so
  // everything that depends on the lane lives here); lanes of query rows beyond the block are switched off

  // ---- register-resident operand R[d][q]: lane (q, kq) keeps rows d = 4 s + kq ----------------
  // fetched by the work-group as 16-byte rows into LDS (the tile buffers are still idle) and picked from there: a
  // wave-level memory instruction costs ~100 cycles whatever its width (see attn_bf16x3.hip).  Chunk c of row r sits at
  // c ^ 4 (r & 1): the lane quarters read rows 1 apart.  Backward: delta_q = sum_d dO[d][q] O[d][q] from a second round with O.
  float R[D / 4];
  float* xbuf = &tiles[0][0][0];
  constexpr int CH_T = D / 16;                                     // 16-byte chunks per thread: D rows x 32 chunks / 512
  const int cc = tid & 31, crow = tid >> 5;
  const unsigned c_off = (qt * 128 + 4 * cc) < Tq_e ? (unsigned)(crow * ld + qt * 128 + 4 * cc) * 4u : CSN_OOB;
  const int col = 16 * wave + lq;
  auto stage_in = [&](const csn_rsrc_t& rs) {
    f32x4 ch[CH_T];
#pragma unroll
    for (int t = 0; t < CH_T; ++t) ch[t] = csn_bload4(rs, c_off, (unsigned)(16 * t * ld) * 4u);
#pragma unroll
    for (int t = 0; t < CH_T; ++t) {
      const int row = crow + 16 * t;
      *reinterpret_cast<f32x4*>(&xbuf[row * 128 + ((cc ^ (4 * (row & 1))) << 2)]) = ch[t];
    }
  };
  auto pick = [&](int row) { return xbuf[row * 128 + ((((col >> 2) ^ (4 * (row & 1))) << 2) | (col & 3))]; };
  stage_in(Rr);
  __syncthreads();
#pragma unroll
  for (int s = 0; s < D / 4; ++s) R[s] = pick(4 * s + kq);
  float delta_q = 0.f;
  if (BWD) {
    const csn_rsrc_t Xr = csn_make_rsrc(p.ctx + qs * p.q_shape_stride + head_off, win);
    __syncthreads();
    stage_in(Xr);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < D / 4; ++s) delta_q = fmaf(R[s], pick(4 * s + kq), delta_q);
    delta_q += __shfl_xor(delta_q, 16, 64);
    delta_q += __shfl_xor(delta_q, 32, 64);
  }
  __syncthreads();                                                 // the staging block becomes the tile buffers

  f32x4v O[D / 16];
#pragma unroll
  for (int c = 0; c < D / 16; ++c) O[c] = f32x4v{0.f, 0.f, 0.f, 0.f};

  // attention-probability dropout (csa_models.py:141): P_drop = mask * P / (1 - p); one hash per key pair (csn_common.h)
  const bool drop = p.dropout_p > 0.f;
  const unsigned thr16 = csn_drop_threshold16(p.dropout_p);
  const float keep_scale = drop ? 1.f / (1.f - p.dropout_p) : 1.f;
  const unsigned salt = csn_block_salt((unsigned long long)(((long long)e * p.H + hd) * p.n_blocks + blk), p.seed);
  const int mp = Tq > Tp ? Tq : Tp;                               // mask pitch: pair index = key pair * mp + query stays unique when n_queries > score_pitch
  const unsigned pw_base = (unsigned)(2 * kq * mp + qrow);       // pair index of this lane's keys 4 kq, 4 kq + 1

  float m_run = -INFINITY, l_run = 0.f;       // forward: running max / partial sum of this lane's key quarter
  float lse_q = 0.f;                           // backward: per-query constant
  if (BWD) {
    lse_q = q_ok ? p.lse[stat_off + qrow] : 0.f;
    if (q_ok && kq == 0 && p.delta) p.delta[stat_off + qrow] = delta_q;
  }

  // ---- streamed tiles: global -> registers -> LDS (swizzled) ---------------------------------------
  // piece idx = tid + 512 i  ->  row idx / 8, keys 4 (idx % 8) .. +3
  const int t_c = (tid & 7) * 4;
  unsigned t_off[NP_T];
  int a_dst[NP_T], b_dst[NP_T];
#pragma unroll
  for (int i = 0; i < NP_T; ++i) {
    const int idx = tid + 512 * i, row = idx >> 3;
    t_off[i] = idx < PIECES ? (unsigned)(row * ldk + t_c) * 4u : CSN_OOB;
    a_dst[i] = row * KT + (t_c ^ (16 * (row & 1)));                    // key half swapped on odd rows
    b_dst[i] = row * KT + 4 * ((tid & 7) ^ ((row >> 1) & 7));          // 16-byte chunk ^ (row >> 1) & 7
  }
  f32x4 g[NP_T];
  auto fetch = [&](const csn_rsrc_t& rs, int kt) {
    const int k0 = kt * KT;
    // pieces past the block end are switched off (a ragged last piece is fetched whole: its extra keys are masked below)
    const unsigned poison = (k0 + t_c) < T ? 0u : CSN_OOB;
#pragma unroll
    for (int i = 0; i < NP_T; ++i) g[i] = csn_bload4(rs, t_off[i] | poison, (unsigned)k0 * 4u);
  };
  auto commitA = [&](int st) {
#pragma unroll
    for (int i = 0; i < NP_T; ++i)
      if (NP_T * 512 == PIECES || tid + 512 * i < PIECES) *reinterpret_cast<f32x4*>(&tileA[st][a_dst[i]]) = g[i];
  };
  auto commitB = [&](int st) {
#pragma unroll
    for (int i = 0; i < NP_T; ++i)
      if (NP_T * 512 == PIECES || tid + 512 * i < PIECES) *reinterpret_cast<f32x4*>(&tileB[st][b_dst[i]]) = g[i];
  };

  // fragment read positions (lane constants)
  const int a_col0 = (16 * (0 ^ (kq & 1))) + lq, a_col1 = (16 * (1 ^ (kq & 1))) + lq;      // tileA, key tiles j = 0, 1
  const int b_ch0 = 4 * ((0 + kq) ^ ((lq >> 1) & 7)), b_ch1 = 4 * ((4 + kq) ^ ((lq >> 1) & 7));  // tileB chunks

  const int nkt = (T + KT - 1) / KT;
  fetch(Ar, 0); commitA(0);
  fetch(Br, 0); commitB(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1, nxt = cur ^ 1;
    const bool more = kt + 1 < nkt;
    if (more) fetch(Ar, kt + 1);

    // score positions of this lane: tile j, reg r  ->  key kt*32 + 16 j + 4 kq + r
    // scores of a block are stored [query][key] (pitch Tp): this lane's keys kt*32 + 16 j + 4 kq .. + 3 are one 16-byte run
    unsigned s_off[8];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * KT + 16 * j + 4 * kq + r;
        s_off[4 * j + r] = (q_ok && key < T) ? (unsigned)(qrow * Tp + key) * 4u : CSN_OOB;
      }
    float sv[8];
    if (BWD) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {                                    // saved scores, requested early
        const f32x4 v = csn_bload4(Sr, s_off[4 * j]);
        sv[4 * j] = v[0]; sv[4 * j + 1] = v[1]; sv[4 * j + 2] = v[2]; sv[4 * j + 3] = v[3];
      }
    }

    // ---- phase 1: T1[key][q] = sum_d tileA[d][key] R[d][q] ------------------------------------
    f32x4v S0 = {0.f, 0.f, 0.f, 0.f}, S1 = {0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ tA = &tileA[cur][kq * KT];
#pragma unroll
    for (int s = 0; s < D / 4; ++s) {
      S0 = mfma16(tA[4 * s * KT + a_col0], R[s], S0);
      S1 = mfma16(tA[4 * s * KT + a_col1], R[s], S1);
    }

    if (more) { commitA(nxt); fetch(Br, kt + 1); }

    // ---- pointwise ----------------------------------------------------------------------------
    float t1[8] = {S0[0], S0[1], S0[2], S0[3], S1[0], S1[1], S1[2], S1[3]};
    bool keep[8];
    if (drop) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const unsigned h = csn_pair_hash(pw_base + (unsigned)((kt * (KT / 2) + 8 * j + w) * mp), salt);
          keep[4 * j + 2 * w] = (h & 0xffffu) >= thr16;
          keep[4 * j + 2 * w + 1] = (h >> 16) >= thr16;
        }
    }
    if (!BWD) {
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int key = kt * KT + 16 * (r >> 2) + 4 * kq + (r & 3);
        if (key >= T) t1[r] = -INFINITY;
        mx = fmaxf(mx, t1[r]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)                    // (zero-sized window when scores are not kept)
        csn_bstore4(f32x4{t1[4 * j], t1[4 * j + 1], t1[4 * j + 2], t1[4 * j + 3]}, Sr, s_off[4 * j]);
      // lazy rescale: only when some query's running maximum would grow by more than the threshold.  The four lanes
      // of a query share m_run, so the cross-lane maximum is only needed inside the (rare) branch.
      if (__any(mx > m_run + p.rescale_threshold)) {
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = (m_new == -INFINITY) ? 1.f : expf(m_run - m_new);
#pragma unroll
        for (int c = 0; c < D / 16; ++c) O[c] *= alpha;
        l_run *= alpha;
        m_run = m_new;
      }
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        t1[r] = expf(t1[r] - m_run);                 // masked keys: exp(-inf) = 0
        ps += t1[r];
      }
      l_run += ps;                                   // the softmax denominator sees every key, dropped or not
      if (drop) {
#pragma unroll
        for (int r = 0; r < 8; ++r) t1[r] = keep[r] ? t1[r] * keep_scale : 0.f;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const bool ok = s_off[r] != CSN_OOB;
        const float pv = ok ? expf(sv[r] - lse_q) : 0.f;           // softmax probability (csa_models.py:141)
        const float md = (!drop || keep[r]) ? keep_scale : 0.f;    // d P_drop / d P
        const float ds = pv * (t1[r] * md - delta_q);              // d softmax (delta = rowsum(dO * O) already has the mask)
        sv[r] = pv * md;                                           // what the dV product needs: the dropped probabilities
        t1[r] = ds;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        csn_bstore4(f32x4{sv[4 * j], sv[4 * j + 1], sv[4 * j + 2], sv[4 * j + 3]}, Sr, s_off[4 * j]);
        csn_bstore4(f32x4{t1[4 * j], t1[4 * j + 1], t1[4 * j + 2], t1[4 * j + 3]}, dSr, s_off[4 * j]);
      }
    }

    // ---- phase 2: OUT[c][q] += sum_key tileB[c][key] T1[key][q] --------------------------------
    const float* __restrict__ tB = &tileB[cur][lq * KT];
#pragma unroll
    for (int c = 0; c < D / 16; ++c) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(&tB[c * 16 * KT + b_ch0]);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(&tB[c * 16 * KT + b_ch1]);
#pragma unroll
      for (int r = 0; r < 4; ++r) O[c] = mfma16(a0[r], t1[r], O[c]);
#pragma unroll
      for (int r = 0; r < 4; ++r) O[c] = mfma16(a1[r], t1[4 + r], O[c]);
    }

    if (more) commitB(nxt);
    __syncthreads();
  }

  // ---- epilogue -----------------------------------------------------------------------------------
  float inv = 1.f;
  if (!BWD) {
    float l_tot = l_run + __shfl_xor(l_run, 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
    inv = 1.f / l_tot;
    if (q_ok && kq == 0 && p.lse) p.lse[stat_off + qrow] = m_run + logf(l_tot);
  }
  // OUT leaves through the same [D][128] LDS block as 16-byte rows (chunk c of row r at c ^ 4 ((r >> 2) & 1)); when several
  // evaluations share the output slot, the previous partial sums are fetched first — one batch of loads — then added
#pragma unroll
  for (int c = 0; c < D / 16; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * c + 4 * kq + r;
      xbuf[row * 128 + ((((col >> 2) ^ (4 * ((row >> 2) & 1))) << 2) | (col & 3))] = O[c][r] * inv;
    }
  __syncthreads();
  {
    f32x4 ch[CH_T];
#pragma unroll
    for (int t = 0; t < CH_T; ++t) {
      const int row = crow + 16 * t;
      ch[t] = *reinterpret_cast<const f32x4*>(&xbuf[row * 128 + ((cc ^ (4 * ((row >> 2) & 1))) << 2)]);
    }
    if (p.accumulate) {
      f32x4 prev[CH_T];
#pragma unroll
      for (int t = 0; t < CH_T; ++t) prev[t] = csn_bload4(Or, c_off, (unsigned)(16 * t * ld) * 4u);
#pragma unroll
      for (int t = 0; t < CH_T; ++t) ch[t] += prev[t];
    }
#pragma unroll
    for (int t = 0; t < CH_T; ++t) csn_bstore4(ch[t], Or, c_off, (unsigned)(16 * t * ld) * 4u);
  }
}

template <int DT>
int launch_dt(const CsnAttnArgs& a, bool bwd, hipStream_t st) {
  const long long units = (long long)a.n_blocks * a.H * a.E;
  dim3 grid((unsigned)(((units + 7) / 8) * 8 * (((a.Tq > 0 ? a.Tq : a.T) + 127) / 128)));
  if (bwd) hipLaunchKernelGGL((csn_attn_f32_kernel<DT, true>), grid, dim3(512), 0, st, a);
  else hipLaunchKernelGGL((csn_attn_f32_kernel<DT, false>), grid, dim3(512), 0, st, a);
  return (int)hipGetLastError();
}

int launch_any(const CsnAttnArgs& a, int d, bool bwd, hipStream_t st) {
  if (a.E <= 0 || a.n_blocks <= 0) return 0;
  if ((a.ld & 3) || (a.Tp & 3) || (a.ld_kv & 3)) return -2;
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
