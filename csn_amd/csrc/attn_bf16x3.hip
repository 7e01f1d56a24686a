// Fused block-diagonal attention for the CSA layer in the "bf16x3" math mode: every fp32 product is three bf16
// matrix-core products (hi*hi + hi*lo + lo*hi, fp32 accumulate; see gemm_bf16x3.hip for the error model).
//
// Same skeleton, tiling, LDS budget and HBM data flow as attn_f32.hip (read that header first):
//   R      [d][q]   register operand, 16 query points per wave        (fwd: Qs^T   bwd: dO^T)    as bf16 hi/lo fragments
//   tileA  [d][key] 32 keys, k-major: fragments by ds_read_b64_tr_b16  (fwd: K^T    bwd: V^T)     bf16 hi / lo planes
//   tileB  [d][key] 32 keys, key-contiguous: fragments by ds_read_b64  (fwd: V^T    bwd: K^T)     bf16 hi / lo planes
// Matrix instruction: v_mfma_f32_16x16x32_bf16 (A: lane l = A[l & 15][8 (l >> 4) + j], B: lane l = B[8 (l >> 4) + j][l & 15],
// j = 0..7; C: reg r of lane l = C[4 (l >> 4) + r][l & 15]) — the C layout of the 16x16 fp32 shape, so the phase-1
// accumulators of two 16-key tiles are, after the hi/lo split, the 32-key B fragment of phase 2 with the k order
//   element j of lane (q, kq)  <->  key 4 kq + j (j < 4)  |  16 + 4 kq + (j - 4) (j >= 4),
// and the V^T / K^T fragments are fetched in that same order (two 8-byte reads per plane).
// The fp32 tiles coming from HBM are split into their bf16 planes while they are staged into LDS (two planes = the
// bytes of the fp32 tile).  LDS images (conflict-free): tileA swaps the 16-key halves on rows with bit 3 set; tileB
// stores, per row, the 8 keys of each lane quarter kq contiguously ([4kq..4kq+3 | 16+4kq..16+4kq+3], one 16-byte read
// per fragment) and XORs that 16-byte unit index with (-(row >> 2)) & 3.
#include "csn_common.h"
#include "csn_kernels.h"

namespace {

constexpr int KT = 32;               // keys per streamed tile

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;

CSN_DEVINL f32x4v mfma16(bf16x8 a, bf16x8 b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// acc += a * b with a = ah + al, b = bh + bl (small terms first)
CSN_DEVINL f32x4v mfma3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x4v c) {
  c = mfma16(al, bh, c);
  c = mfma16(ah, bl, c);
  return mfma16(ah, bh, c);
}

CSN_DEVINL bf16x8 join8(s16x4 a, s16x4 b) {
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

CSN_DEVINL void split4(const f32x4 v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hi[i] = (__bf16)v[i];
    lo[i] = (__bf16)(v[i] - (float)hi[i]);
  }
}

template <int DT, bool BWD>
__global__ __launch_bounds__(512, 2) void csn_attn_bf16x3_kernel(CsnAttnArgs p) {
  constexpr int D = 32 * DT;
  constexpr int PIECES = D * 8;                         // 16-byte pieces per streamed tile
  constexpr int NP_T = (PIECES + 511) / 512;            // pieces per thread per tile
  // [stage][plane hi/lo][row][32 keys]
  __shared__ __attribute__((aligned(16))) __bf16 tileA[2][2][D * KT];
  __shared__ __attribute__((aligned(16))) __bf16 tileB[2][2][D * KT];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 15, kq = lane >> 4;
  // XCD-aware work-group order.  A unit = one (evaluation, head, block); its QT query tiles all stream the same K/V
  // block, so they should share one XCD's L2.  Work-groups are dealt round-robin over the 8 XCDs, hence the QT tiles of
  // unit u get ids 8 * (QT * (u / 8) + qt) + (u % 8): same residue mod 8 (same XCD), adjacent in dispatch order.
  // (Placement only changes speed: every tile is self-contained.)
  const int QT = (p.T + 127) / 128;
  const int Y = p.n_blocks * p.H;
  const int L = blockIdx.x, slot = L & 7, jj = L >> 3;
  const int qt = jj % QT, u = (jj / QT) * 8 + slot;
  if (u >= Y * p.E) return;
  const int e = p.eval_ids ? p.eval_ids[u / Y] : u / Y;
  const int hd = (u % Y) % p.H, blk = (u % Y) / p.H;
  const int T = p.T, Tp = p.Tp, ld = p.ld;
  const int qrow = qt * 128 + wave * 16 + lq;                  // query index inside the block
  const bool q_ok = qrow < T;

  const long long qs = p.q_index ? p.q_index[e] : e;
  const long long ks = p.kv_index ? p.kv_index[e] : e;
  const long long os = p.out_index ? p.out_index[e] : e;
  const long long head_off = (long long)hd * D * ld + (long long)blk * T;
  const long long win = ((long long)(D - 1) * ld + T) * 4;     // bytes spanned by a [D][T] window of pitch ld
  const csn_rsrc_t Rr = csn_make_rsrc(p.q + qs * p.q_shape_stride + head_off, win);
  const csn_rsrc_t Ar = csn_make_rsrc((BWD ? p.v : p.k) + ks * p.kv_shape_stride + head_off, win);
  const csn_rsrc_t Br = csn_make_rsrc((BWD ? p.k : p.v) + ks * p.kv_shape_stride + head_off, win);
  const csn_rsrc_t Or = csn_make_rsrc(p.out + os * p.out_eval_stride + head_off, win);
  const long long stat_off = ((long long)e * p.H + hd) * ((long long)p.n_blocks * T) + (long long)blk * T;
  const long long sc_off = (((long long)e * p.H + hd) * p.n_blocks + blk) * ((long long)T * Tp);
  const bool have_scores = p.scores != nullptr;
  const csn_rsrc_t Sr = csn_make_rsrc(have_scores ? p.scores + sc_off : nullptr, have_scores ? (long long)T * Tp * 4 : 0);
  const csn_rsrc_t dSr = csn_make_rsrc(BWD ? p.dscores + sc_off : nullptr, BWD ? (long long)T * Tp * 4 : 0);

  // per-lane byte offsets (scalar offsets handed to the buffer instructions must be wave-uniform, so
  // everything that depends on the lane lives here); lanes of query rows beyond the block are switched off
  const unsigned r_off = q_ok ? (unsigned)(8 * kq * ld + qrow) * 4u : CSN_OOB;    // rows 32 s + 8 kq + j
  const unsigned o_off = q_ok ? (unsigned)(4 * kq * ld + qrow) * 4u : CSN_OOB;    // rows 16 c + 4 kq + r

  // ---- register-resident operand R[d][q]: lane (q, kq) keeps rows d = 32 s + 8 kq + j as bf16 hi / lo ------
  bf16x8 Rh[D / 32], Rl[D / 32];
#pragma unroll
  for (int s = 0; s < D / 32; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = csn_bload(Rr, r_off, (unsigned)(32 * s + j) * ld * 4u);
      Rh[s][j] = (__bf16)v;
      Rl[s][j] = (__bf16)(v - (float)Rh[s][j]);
    }

  f32x4v O[D / 16];
#pragma unroll
  for (int c = 0; c < D / 16; ++c) O[c] = f32x4v{0.f, 0.f, 0.f, 0.f};

  // attention-probability dropout (csa_models.py:141): P_drop = mask * P / (1 - p); element index = position in `scores`
  const bool drop = p.dropout_p > 0.f;
  const unsigned thr24 = csn_drop_threshold(p.dropout_p);
  const float keep_scale = drop ? 1.f / (1.f - p.dropout_p) : 1.f;

  float m_run = -INFINITY, l_run = 0.f;       // forward: running max / partial sum of this lane's key quarter
  float lse_q = 0.f, delta_q = 0.f;           // backward: per-query constants
  if (BWD) {
    lse_q = q_ok ? p.lse[stat_off + qrow] : 0.f;
    delta_q = q_ok ? p.delta[stat_off + qrow] : 0.f;
  }

  // ---- streamed tiles: global -> registers -> LDS (swizzled) ---------------------------------------
  // piece idx = tid + 512 i  ->  row idx / 8, keys 4 (idx % 8) .. +3
  const int t_c = (tid & 7) * 4;
  unsigned t_off[NP_T];
  int a_dst[NP_T], b_dst[NP_T];
#pragma unroll
  for (int i = 0; i < NP_T; ++i) {
    const int idx = tid + 512 * i, row = idx >> 3;
    t_off[i] = idx < PIECES ? (unsigned)(row * ld + t_c) * 4u : CSN_OOB;
    a_dst[i] = row * KT + (t_c ^ (16 * ((row >> 3) & 1)));             // key halves swapped on rows with bit 3 set
    // tileB: 4-key chunk c8 -> 8-byte slot s8 = 2 c8 (keys 0..15) | 2 (c8 - 4) + 1 (keys 16..31); its 16-byte unit
    // (s8 >> 1 = kq) is XORed with (-(row >> 2)) & 3
    const int c8 = tid & 7, s8 = c8 < 4 ? 2 * c8 : 2 * (c8 - 4) + 1;
    b_dst[i] = row * KT + 4 * (2 * ((s8 >> 1) ^ ((-((row >> 2) & 3)) & 3)) + (s8 & 1));
  }
  f32x4 g[NP_T];
  auto fetch = [&](const csn_rsrc_t& rs, int kt) {
    const int k0 = kt * KT;
    // T % 4 == 0: a 16-byte piece is all in or all out; pieces past the block end are switched off
    const unsigned poison = (k0 + t_c) < T ? 0u : CSN_OOB;
#pragma unroll
    for (int i = 0; i < NP_T; ++i) g[i] = csn_bload4(rs, t_off[i] | poison, (unsigned)k0 * 4u);
  };
  auto commitA = [&](int st) {
#pragma unroll
    for (int i = 0; i < NP_T; ++i)
      if (NP_T * 512 == PIECES || tid + 512 * i < PIECES) {
        bf16x4 hi, lo;
        split4(g[i], hi, lo);
        *reinterpret_cast<bf16x4*>(&tileA[st][0][a_dst[i]]) = hi;
        *reinterpret_cast<bf16x4*>(&tileA[st][1][a_dst[i]]) = lo;
      }
  };
  auto commitB = [&](int st) {
#pragma unroll
    for (int i = 0; i < NP_T; ++i)
      if (NP_T * 512 == PIECES || tid + 512 * i < PIECES) {
        bf16x4 hi, lo;
        split4(g[i], hi, lo);
        *reinterpret_cast<bf16x4*>(&tileB[st][0][b_dst[i]]) = hi;
        *reinterpret_cast<bf16x4*>(&tileB[st][1][b_dst[i]]) = lo;
      }
  };

  // fragment read positions (lane constants).  tileA, transposing read: inside a 16-lane group lane 4 q' + p addresses
  // row 8 kq + q', keys 4 p .. 4 p + 3 of the 16-key tile t (halves swapped when (row >> 3) & 1 = kq & 1 is set)
  const int tr_row = 8 * kq + (lq >> 2);
  const int a_pos0 = tr_row * KT + 16 * (0 ^ (kq & 1)) + 4 * (lq & 3), a_pos1 = tr_row * KT + 16 * (1 ^ (kq & 1)) + 4 * (lq & 3);
  // tileB: row lq of the 16-channel tile, 16-byte unit kq ^ ((-(lq >> 2)) & 3): keys 4kq..4kq+3 then 16+4kq..16+4kq+3
  const int b_pos = lq * KT + 8 * (kq ^ ((-((lq >> 2) & 3)) & 3));

  const int nkt = (T + KT - 1) / KT;
  fetch(Ar, 0); commitA(0);
  fetch(Br, 0); commitB(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1, nxt = cur ^ 1;
    const bool more = kt + 1 < nkt;
    if (more) fetch(Ar, kt + 1);

    // score positions of this lane: tile j, reg r  ->  key kt*32 + 16 j + 4 kq + r
    unsigned s_off[8];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * KT + 16 * j + 4 * kq + r;
        s_off[4 * j + r] = (q_ok && key < T) ? (unsigned)(key * Tp + qrow) * 4u : CSN_OOB;
      }
    float sv[8];
    if (BWD) {
#pragma unroll
      for (int r = 0; r < 8; ++r) sv[r] = csn_bload(Sr, s_off[r]);     // saved scores, requested early
    }

    // ---- phase 1: T1[key][q] = sum_d tileA[d][key] R[d][q] ------------------------------------
    f32x4v S0 = {0.f, 0.f, 0.f, 0.f}, S1 = {0.f, 0.f, 0.f, 0.f};
    const __bf16* __restrict__ tAh = tileA[cur][0];
    const __bf16* __restrict__ tAl = tileA[cur][1];
#pragma unroll
    for (int s = 0; s < D / 32; ++s) {
      const int o = 32 * s * KT;
      const bf16x8 a0h = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tAh + o + a_pos0)),
                               __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tAh + o + a_pos0 + 4 * KT)));
      const bf16x8 a0l = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tAl + o + a_pos0)),
                               __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tAl + o + a_pos0 + 4 * KT)));
      const bf16x8 a1h = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tAh + o + a_pos1)),
                               __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tAh + o + a_pos1 + 4 * KT)));
      const bf16x8 a1l = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tAl + o + a_pos1)),
                               __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tAl + o + a_pos1 + 4 * KT)));
      S0 = mfma3(a0h, a0l, Rh[s], Rl[s], S0);
      S1 = mfma3(a1h, a1l, Rh[s], Rl[s], S1);
    }

    if (more) { commitA(nxt); fetch(Br, kt + 1); }

    // ---- pointwise ----------------------------------------------------------------------------
    float t1[8] = {S0[0], S0[1], S0[2], S0[3], S1[0], S1[1], S1[2], S1[3]};
    if (!BWD) {
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int key = kt * KT + 16 * (r >> 2) + 4 * kq + (r & 3);
        if (key >= T) t1[r] = -INFINITY;
        csn_bstore(t1[r], Sr, s_off[r]);             // (zero-sized window when scores are not kept)
        mx = fmaxf(mx, t1[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      // lazy rescale: only when some query's running maximum would grow by more than the threshold
      if (__any(mx > m_run + p.rescale_threshold)) {
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
        t1[r] = __expf(t1[r] - m_run);               // hardware exp2 path (~2 ulp: far below the bf16x3 error); exp(-inf) = 0
        ps += t1[r];
      }
      l_run += ps;                                   // the softmax denominator sees every key, dropped or not
      if (drop) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int key = kt * KT + 16 * (r >> 2) + 4 * kq + (r & 3);
          t1[r] = csn_keep((unsigned long long)(sc_off + (long long)key * Tp + qrow), p.seed, thr24) ? t1[r] * keep_scale : 0.f;
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const bool ok = s_off[r] != CSN_OOB;
        const float pv = ok ? __expf(sv[r] - lse_q) : 0.f;         // softmax probability (csa_models.py:141)
        float md = 1.f;                                            // d P_drop / d P
        if (drop) {
          const int key = kt * KT + 16 * (r >> 2) + 4 * kq + (r & 3);
          md = csn_keep((unsigned long long)(sc_off + (long long)key * Tp + qrow), p.seed, thr24) ? keep_scale : 0.f;
        }
        const float ds = pv * (t1[r] * md - delta_q);              // d softmax (delta = rowsum(dO * O) already has the mask)
        csn_bstore(pv * md, Sr, s_off[r]);                         // what the dV product needs: the dropped probabilities
        csn_bstore(ds, dSr, s_off[r]);
        t1[r] = ds;
      }
    }

    // ---- phase 2: OUT[c][q] += sum_key tileB[c][key] T1[key][q] --------------------------------
    bf16x8 ph, pl;                                      // T1 as a 32-key B fragment, split into bf16 hi / lo
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      ph[r] = (__bf16)t1[r];
      pl[r] = (__bf16)(t1[r] - (float)ph[r]);
    }
    const __bf16* __restrict__ tBh = tileB[cur][0];
    const __bf16* __restrict__ tBl = tileB[cur][1];
#pragma unroll
    for (int c = 0; c < D / 16; ++c) {
      const int o = c * 16 * KT;
      const bf16x8 vh = *reinterpret_cast<const bf16x8*>(tBh + o + b_pos);
      const bf16x8 vl = *reinterpret_cast<const bf16x8*>(tBl + o + b_pos);
      O[c] = mfma3(vh, vl, ph, pl, O[c]);
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
  if (p.accumulate) {
    // several evaluations share this output slot: fetch all previous partial sums first (one batch of loads in
    // flight), then add and store — a load/add/store chain per element would serialise 64 memory round trips
    f32x4v prev[D / 16];
#pragma unroll
    for (int c = 0; c < D / 16; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) prev[c][r] = csn_bload(Or, o_off, (unsigned)(c * 16 + r) * ld * 4u);
#pragma unroll
    for (int c = 0; c < D / 16; ++c) O[c] = O[c] * inv + prev[c];
  } else {
#pragma unroll
    for (int c = 0; c < D / 16; ++c) O[c] *= inv;
  }
#pragma unroll
  for (int c = 0; c < D / 16; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) csn_bstore(O[c][r], Or, o_off, (unsigned)(c * 16 + r) * ld * 4u);
}

template <int DT>
int launch_dt(const CsnAttnArgs& a, bool bwd, hipStream_t st) {
  const long long units = (long long)a.n_blocks * a.H * a.E;
  dim3 grid((unsigned)(((units + 7) / 8) * 8 * ((a.T + 127) / 128)));
  if (bwd) hipLaunchKernelGGL((csn_attn_bf16x3_kernel<DT, true>), grid, dim3(512), 0, st, a);
  else hipLaunchKernelGGL((csn_attn_bf16x3_kernel<DT, false>), grid, dim3(512), 0, st, a);
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

int csn_launch_attn_fwd_bf16x3(const CsnAttnArgs& a, int d, hipStream_t st) { return launch_any(a, d, false, st); }
int csn_launch_attn_bwd_bf16x3(const CsnAttnArgs& a, int d, hipStream_t st) { return launch_any(a, d, true, st); }
