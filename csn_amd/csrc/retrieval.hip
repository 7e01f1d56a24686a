// Shape-retrieval measure for the kNN shape graph (MID-FC/csa_models.py:244-267):
//     r[i][j] = mean_n max_m cos(f1[i][n][:], f2[j][m][:])
// The reference materialises the N x N cosine matrix of every (query, candidate) pair (400 MB at
// N = 10^4); here it never exists: a work-group owns 128 query points of one pair, sweeps the
// candidate's points 128 at a time through the fp32 matrix cores and keeps only the running maximum.
//
// Features are POINT-MAJOR [shape][point][channel] here (that is how get_all_feats hands them over,
// csa_models.py:299), so both operands are "k contiguous": G^T[m][n] = sum_c f2[m][c] f1[n][c] with the
// query point n on the lanes and the candidate point m on the accumulator registers — the maximum over
// m is then a register/half/tile reduction with no cross-lane traffic until the very end.
#include "csn_common.h"
#include "csn_kernels.h"

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;

// inv[r] = 1 / max(||f[r][:]||_2, eps)      one wave per row (F.normalize, eps = 1e-12)
__global__ __launch_bounds__(256) void csn_row_inv_norm_kernel(const float* __restrict__ f, float* __restrict__ inv,
                                                               long long rows, int C, float eps) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* __restrict__ p = f + row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += p[c] * p[c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) inv[row] = 1.f / fmaxf(sqrtf(s), eps);
}

// rowmax[(i*s2 + j)][n] = max_m inv1[i][n] inv2[j][m] <f1[i][n], f2[j][m]>
__global__ __launch_bounds__(256, 2) void csn_retrieval_rowmax_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                                      const float* __restrict__ inv1, const float* __restrict__ inv2,
                                                                      float* __restrict__ rowmax, int s2, int n1, int n2, int C) {
  __shared__ __attribute__((aligned(16))) float As[128 * LDK];      // candidate points (rows m)
  __shared__ __attribute__((aligned(16))) float Bs[128 * LDK];      // query points (rows n)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  // 1-D grid, pair-major: blockIdx.x = pair * tiles + tile (a y dimension would cap the pair count at 65535)
  const int tiles = (n1 + 127) / 128;
  const int pair = blockIdx.x / tiles, i = pair / s2, j = pair % s2;
  const int nq0 = (blockIdx.x % tiles) * 128;

  const csn_rsrc_t Qr = csn_make_rsrc(f1 + ((long long)i * n1 + nq0) * C, (long long)min(128, n1 - nq0) * C * 4);
  const int pr = tid >> 3, pc = (tid & 7) * 4;
  unsigned q_off[4], c_off[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    q_off[a] = (unsigned)((pr + 32 * a) * C + pc) * 4u;     // rows past n1 fall outside the window -> 0
    c_off[a] = (unsigned)((pr + 32 * a) * C + pc) * 4u;
  }
  float qinv[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int n = nq0 + wn0 + 32 * b + l31;
    qinv[b] = n < n1 ? inv1[(long long)i * n1 + n] : 0.f;
  }
  float best[2] = {-INFINITY, -INFINITY};

  for (int m0 = 0; m0 < n2; m0 += 128) {
    const csn_rsrc_t Cr = csn_make_rsrc(f2 + ((long long)j * n2 + m0) * C, (long long)min(128, n2 - m0) * C * 4);
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    for (int k0 = 0; k0 < C; k0 += BK) {
      const unsigned kp = (k0 + pc) < C ? 0u : CSN_OOB;
      f32x4 ra[4], rb[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        ra[a] = csn_bload4(Cr, c_off[a] | kp, (unsigned)k0 * 4u);
        rb[a] = csn_bload4(Qr, q_off[a] | kp, (unsigned)k0 * 4u);
      }
      __syncthreads();                                   // previous slab fully consumed
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        *reinterpret_cast<f32x4*>(&As[(pr + 32 * a) * LDK + pc]) = ra[a];
        *reinterpret_cast<f32x4*>(&Bs[(pr + 32 * a) * LDK + pc]) = rb[a];
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < BK; kk += 8) {
        f32x4 af[2], bf[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) af[a] = *reinterpret_cast<const f32x4*>(&As[(wm0 + 32 * a + l31) * LDK + kk + 4 * h]);
#pragma unroll
        for (int b = 0; b < 2; ++b) bf[b] = *reinterpret_cast<const f32x4*>(&Bs[(wn0 + 32 * b + l31) * LDK + kk + 4 * h]);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = csn_mfma(af[a][t], bf[b][t], acc[a][b]);
      }
    }
    // fold this 128-candidate tile into the running maxima (candidate m on registers, query n on lanes)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + 32 * a + csn_acc_row(r, h);
        const float im = m < n2 ? inv2[(long long)j * n2 + m] : 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const float cosv = acc[a][b][r] * qinv[b] * im;
          if (m < n2) best[b] = fmaxf(best[b], cosv);
        }
      }
  }
  // combine the two lane halves, then the two wave rows (wm0 = 0 / 64) through LDS
  __syncthreads();
  float* red = As;                                        // [2 wave rows][128 n]
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    best[b] = fmaxf(best[b], csn_xhalf(best[b]));
    if (h == 0) red[(wave >> 1) * 128 + wn0 + 32 * b + l31] = best[b];
  }
  __syncthreads();
  if (tid < 128) {
    const int n = nq0 + tid;
    if (n < n1) rowmax[(long long)pair * n1 + n] = fmaxf(red[tid], red[128 + tid]);
  }
}

// out[pair] = (1/n) * sum_n rowmax[pair][n], fixed-order tree -> bitwise reproducible
__global__ __launch_bounds__(256) void csn_row_mean_kernel(const float* __restrict__ rowmax, float* __restrict__ out, int n) {
  __shared__ float red[256];
  const float* __restrict__ p = rowmax + (long long)blockIdx.x * n;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += p[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0] / (float)n;
}

}  // namespace

int csn_launch_retrieval_f32(const float* f1, const float* f2, float* out, int s1, int n1, int s2, int n2, int C,
                             float* ws, hipStream_t st) {
  float* inv1 = ws;
  float* inv2 = ws + (long long)s1 * n1;
  float* rowmax = inv2 + (long long)s2 * n2;
  const long long r1 = (long long)s1 * n1, r2 = (long long)s2 * n2;
  hipLaunchKernelGGL(csn_row_inv_norm_kernel, dim3((unsigned)((r1 + 3) / 4)), dim3(256), 0, st, f1, inv1, r1, C, 1e-12f);
  hipLaunchKernelGGL(csn_row_inv_norm_kernel, dim3((unsigned)((r2 + 3) / 4)), dim3(256), 0, st, f2, inv2, r2, C, 1e-12f);
  const long long blocks = (long long)((n1 + 127) / 128) * s1 * s2;
  if (blocks > 0x7fffffffLL) return -1;                    // CSN_E_ARG: score fewer query shapes per call
  hipLaunchKernelGGL(csn_retrieval_rowmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, f1, f2, inv1, inv2,
                     rowmax, s2, n1, n2, C);
  hipLaunchKernelGGL(csn_row_mean_kernel, dim3(s1 * s2), dim3(256), 0, st, rowmax, out, n1);
  return (int)hipGetLastError();
}
