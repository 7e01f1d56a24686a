// Batched exact-fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   C[z][M][N] = alpha * A[z][M][K] * B[z]           (row-major C, leading dimension ldc)
//     A : "MK"  row-major, the contraction index k is contiguous (16-byte LDS fragment reads)
//     B : "KN"  row-major K x N (k-major, lanes run along n)          -> B_NK = false
//         "NK"  row-major N x K (k contiguous for every output column) -> B_NK = true
//
// This one kernel serves every plain contraction of the cross-shape-attention path; with all
// activations channel-major ([channel][point]) no operand ever needs a transpose:
//   projections        Q^T/K^T/V^T[d][n] = W[d][c] x[c][n]            (csa_models.py:103-105)  A=W     B=x (KN)
//   dCtx^T[D][n]       = W_fc^T[D][c] dZ^T[c][n]                      (backward of :115)       A=Wfc^T B=dZ^T (KN)
//   dW_fc[c][D]        = sum_n dZ^T[c][n] Ctx^T[D][n]                                          A=dZ^T  B=Ctx^T (NK), split over n
//   dW_{q,k,v}[d][c]   = sum_n dQ^T[d][n] x[c][n]                                              A=dQ^T  B=x (NK), split over n
//   dV^T[c][key]       = sum_q dO^T[c][q] P^T[key][q]                 (backward of :142)       A=dO^T  B=P^T (NK)
//   dK^T[d][key]       = sum_q Qs^T[d][q] dS^T[key][q]                (backward of :139)       A=Qs^T  B=dS^T (NK)
//
// Work-group: 256 threads = 4 waves in a 2 x 2 grid, each wave owns (BM/2) x (BN/2) of the output as
// (BM/64) x (BN/64) accumulator tiles of 32 x 32.  K is consumed 32 at a time through one LDS buffer;
// the next K slab is prefetched into registers while the matrix cores work on the current one.
// The fp32 matrix instruction issues once per 64 cycles per SIMD, so with >= 2 work-groups per CU the
// loads of one group hide under the MFMAs of the other.
#include "csn_common.h"
#include "csn_kernels.h"

namespace {

constexpr int BK = 32;
constexpr int LDS_PAD = 4;              // 36-float rows: conflict-free ds_read_b128 for 16 distinct rows
constexpr int LDK = BK + LDS_PAD;

template <int BM, int BN, bool B_NK>
__global__ __launch_bounds__(256, 2) void csn_gemm_f32_kernel(CsnGemmArgs p) {
  constexpr int MT = BM / 64, NT = BN / 64;
  constexpr int A_PASS = BM / 32;                      // 16-byte pieces per thread for the A slab
  constexpr int B_PASS = BN / 32;                      // 16-byte pieces per thread for the B slab
  constexpr int TPR = BN / 4;                          // KN: threads per k row
  constexpr int RPP = 256 / TPR;                       // KN: k rows per pass
  constexpr int A_EL = BM * LDK, B_EL = B_NK ? BN * LDK : BK * BN;
  constexpr int STAGE_EL = 4 * 32 * (BN / 2);                          // epilogue: 32 rows x (BN/2) columns per wave
  __shared__ __attribute__((aligned(16))) float smem[(A_EL + B_EL) > STAGE_EL ? (A_EL + B_EL) : STAGE_EL];
  float* As = smem;
  float* Bs = smem + A_EL;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);

  // XCD-aware work-group order: the tiles of one batch item z share its A and B panels, so they should share one
  // XCD's L2.  Work-groups are dealt round-robin over the 8 XCDs; tile t of item z gets id 8 * (tiles * (z / 8) + t) + z % 8.
  const int tiles_n = (p.N + BN - 1) / BN, tiles = tiles_n * ((p.M + BM - 1) / BM);
  const int L = blockIdx.x, jj = L >> 3;
  const int tile = jj % tiles;
  int z = (jj / tiles) * 8 + (L & 7);
  if (z >= p.batch) return;
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  const int z0 = z % p.n0; z /= p.n0;
  const int z1 = z % p.n1;
  const int z2 = p.eval_ids ? p.eval_ids[z / p.n1] : z / p.n1;
  const int lda = p.A.ld, ldb = p.B.ld, ldc = p.C.ld;
  const int M = p.M, N = (p.n_last > 0 && z0 == p.n0 - 1) ? p.n_last : (p.n_arr ? ((p.n_arr[z2] + 3) & ~3) : p.N);   // ragged batches: this item's own columns (rounded up to 4) / contraction count
  int K = (p.k_last > 0 && z0 == p.n0 - 1) ? p.k_last : (p.k_arr ? p.k_arr[z2] : p.K);
  if (p.k_chunk > 0) { K = min(p.k_chunk, p.K - z0 * p.k_chunk); }
  if (tile_n * BN >= N) return;                                // (a tile beyond a short item's columns; before any barrier)                                // (a tile beyond a short item's columns; before any barrier)
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // wave-uniform windows: the BM rows of A, the B slab columns/rows, the BM x BN tile of C
  const csn_rsrc_t Ar = csn_make_rsrc(csn_operand_base(p.A, z0, z1, z2) + (long long)m0 * lda, (long long)BM * lda * 4);
  const csn_rsrc_t Br = B_NK ? csn_make_rsrc(csn_operand_base(p.B, z0, z1, z2) + (long long)n0 * ldb, (long long)BN * ldb * 4)
                             : csn_make_rsrc(csn_operand_base(p.B, z0, z1, z2) + n0, ((long long)(K - 1) * ldb + (N - n0)) * 4);
  const csn_rsrc_t Cr = csn_make_rsrc(csn_operand_base(p.C, z0, z1, z2) + (long long)m0 * ldc + n0, (long long)BM * ldc * 4);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // per-thread piece coordinates and byte offsets (rows outside the matrix are switched off)
  const int pr = tid >> 3, pc = (tid & 7) * 4;           // MK / NK slabs: row, first k of the piece
  const int kr = tid / TPR, kc = (tid % TPR) * 4;        // KN slab: k row, first column of the piece
  unsigned a_off[A_PASS], b_off[B_PASS];
#pragma unroll
  for (int i = 0; i < A_PASS; ++i) a_off[i] = (m0 + pr + 32 * i) < M ? (unsigned)((pr + 32 * i) * lda + pc) * 4u : CSN_OOB;
#pragma unroll
  for (int i = 0; i < B_PASS; ++i) {
    if (B_NK) b_off[i] = (n0 + pr + 32 * i) < N ? (unsigned)((pr + 32 * i) * ldb + pc) * 4u : CSN_OOB;
    else b_off[i] = (n0 + kc) < N ? (unsigned)((kr + RPP * i) * ldb + kc) * 4u : CSN_OOB;
  }

  f32x4 ra[A_PASS], rb[B_PASS];
  auto load_slab = [&](int k0) {
    const unsigned kp = (k0 + pc) < K ? 0u : CSN_OOB;      // K % 4 == 0: pieces are all in or all out
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) ra[i] = csn_bload4(Ar, a_off[i] | kp, (unsigned)k0 * 4u);
    if (B_NK) {
#pragma unroll
      for (int i = 0; i < B_PASS; ++i) rb[i] = csn_bload4(Br, b_off[i] | kp, (unsigned)k0 * 4u);
    } else {
#pragma unroll
      for (int i = 0; i < B_PASS; ++i) {
        const unsigned kq = (k0 + kr + RPP * i) < K ? 0u : CSN_OOB;
        rb[i] = csn_bload4(Br, b_off[i] | kq, (unsigned)k0 * (unsigned)ldb * 4u);
      }
    }
  };
  auto store_slab = [&]() {
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) *reinterpret_cast<f32x4*>(&As[(pr + 32 * i) * LDK + pc]) = ra[i];
    if (B_NK) {
#pragma unroll
      for (int i = 0; i < B_PASS; ++i) *reinterpret_cast<f32x4*>(&Bs[(pr + 32 * i) * LDK + pc]) = rb[i];
    } else {
#pragma unroll
      for (int i = 0; i < B_PASS; ++i) *reinterpret_cast<f32x4*>(&Bs[(kr + RPP * i) * BN + kc]) = rb[i];
    }
  };

  const int nk = (K + BK - 1) / BK;
  if (nk > 0) { load_slab(0); store_slab(); }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_slab((kt + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 8) {
      f32x4 af[MT], bf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i)
        af[i] = *reinterpret_cast<const f32x4*>(&As[(wm0 + 32 * i + l31) * LDK + kk + 4 * h]);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (B_NK) {
          bf[j] = *reinterpret_cast<const f32x4*>(&Bs[(wn0 + 32 * j + l31) * LDK + kk + 4 * h]);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) bf[j][t] = Bs[(kk + 4 * h + t) * BN + wn0 + 32 * j + l31];
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = csn_mfma(af[i][t], bf[j][t], acc[i][j]);
    }
    __syncthreads();
    if (kt + 1 < nk) { store_slab(); __syncthreads(); }
  }

  // ---- epilogue: every wave transposes its block through LDS (32 rows at a time: the slab buffers are idle now) and moves
  // 16 contiguous bytes per lane — a wave-level memory instruction costs ~100 cycles whatever its width, and the
  // accumulator layout would need 4x as many of them (see gemm_bf16x3.hip)
  const float alpha = p.alpha;
  constexpr int WN = BN / 2;                                           // columns of a wave's block (64)
  constexpr int CPR = WN / 4;                                          // 16-byte chunks per row (16)
  constexpr int RPI = 64 / CPR;                                        // rows covered by one wave-wide chunk access (4)
  constexpr int NCH = 32 / RPI;                                        // chunk accesses per 32-row pass (8)
  float* wbuf = smem + wave * 32 * WN;
  const int cc = lane % CPR, rsub = lane / CPR;
  const int col = wn0 + 4 * cc;
  const bool n_ok = (n0 + col) < N;                                    // N % 4 == 0: a chunk is all in or all out
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r] * alpha;
        if ((m0 + wm0 + 32 * i + csn_acc_row(r, h)) < p.div_rows) v = v / p.div_val;     // q / temperature (csa_models.py:139)
        wbuf[csn_acc_row(r, h) * WN + 32 * j + l31] = v;
      }
    f32x4 vals[NCH];
    unsigned off[NCH];
#pragma unroll
    for (int t = 0; t < NCH; ++t) {
      const int row = RPI * t + rsub;
      vals[t] = *reinterpret_cast<const f32x4*>(&wbuf[row * WN + 4 * cc]);
      const int ml = wm0 + 32 * i + row;
      off[t] = (n_ok && (m0 + ml) < M) ? (unsigned)(ml * ldc + col) * 4u : CSN_OOB;
    }
    if (p.accumulate) {                                                // all loads in flight first, then add + store
      f32x4 prev[NCH];
#pragma unroll
      for (int t = 0; t < NCH; ++t) prev[t] = csn_bload4(Cr, off[t]);
#pragma unroll
      for (int t = 0; t < NCH; ++t) vals[t] += prev[t];
    }
#pragma unroll
    for (int t = 0; t < NCH; ++t) csn_bstore4(vals[t], Cr, off[t]);
  }
}

// out[i] = alpha * sum_z slab[z][i]  (+ out[i] if accumulate) — closes a split-K weight gradient.
__global__ __launch_bounds__(256) void csn_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                              int n_slabs, long long n, float alpha, int accumulate) {
  // 16 bytes per lane, four slabs in flight per accumulator chain (a dword-per-lane loop with one chain ran at 1.9 TB/s)
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 3 < n && (n & 3) == 0) {
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int z = 0;
    for (; z + 3 < n_slabs; z += 4) {
      s0 += *reinterpret_cast<const f32x4*>(slab + (long long)z * n + i);
      s1 += *reinterpret_cast<const f32x4*>(slab + (long long)(z + 1) * n + i);
      s2 += *reinterpret_cast<const f32x4*>(slab + (long long)(z + 2) * n + i);
      s3 += *reinterpret_cast<const f32x4*>(slab + (long long)(z + 3) * n + i);
    }
    for (; z < n_slabs; ++z) s0 += *reinterpret_cast<const f32x4*>(slab + (long long)z * n + i);
    f32x4 s = ((s0 + s1) + (s2 + s3)) * alpha;
    if (accumulate) s += *reinterpret_cast<const f32x4*>(out + i);
    *reinterpret_cast<f32x4*>(out + i) = s;
  } else {
    for (long long j = i; j < n; ++j) {
      float s = 0.f;
      for (int z = 0; z < n_slabs; ++z) s += slab[(long long)z * n + j];
      s *= alpha;
      if (accumulate) s += out[j];
      out[j] = s;
    }
  }
}

}  // namespace

template <int BM, int BN, bool B_NK>
static int launch(const CsnGemmArgs& a, int batch, hipStream_t st) {
  CsnGemmArgs b = a;
  b.batch = batch;
  const long long tiles = (long long)((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM);
  dim3 grid((unsigned)(((batch + 7) / 8) * 8 * tiles));
  hipLaunchKernelGGL((csn_gemm_f32_kernel<BM, BN, B_NK>), grid, dim3(256), 0, st, b);
  return (int)hipGetLastError();
}

int csn_launch_gemm_f32(const CsnGemmArgs& a, int b_is_nk, int batch, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0 || batch <= 0) return 0;
  if ((a.A.ld & 3) || (a.B.ld & 3) || (a.K & 3) || (a.k_chunk & 3)) return -2;   // 16-byte pieces: rows, K multiples of 4
  if (!b_is_nk && (a.N & 3)) return -2;
  if ((reinterpret_cast<uintptr_t>(a.A.ptr) & 15) || (reinterpret_cast<uintptr_t>(a.B.ptr) & 15)) return -3;
  if ((a.A.s0 & 3) || (a.A.s1 & 3) || (a.A.s2 & 3) || (a.B.s0 & 3) || (a.B.s1 & 3) || (a.B.s2 & 3)) return -4;
  if (a.M <= 64) return b_is_nk ? launch<64, 128, true>(a, batch, st) : launch<64, 128, false>(a, batch, st);
  return b_is_nk ? launch<128, 128, true>(a, batch, st) : launch<128, 128, false>(a, batch, st);
}

int csn_launch_slab_reduce(const float* slab, float* out, int n_slabs, long long n, float alpha, int accumulate,
                           hipStream_t st) {
  if (n <= 0) return 0;
  // one wave per work-group: the output is small (a weight matrix), many narrow work-groups keep every CU streaming slabs
  hipLaunchKernelGGL(csn_slab_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(64), 0, st, slab, out, n_slabs, n,
                     alpha, accumulate);
  return (int)hipGetLastError();
}
