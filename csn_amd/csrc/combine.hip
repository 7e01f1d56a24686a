// Cross-shape mixing stage of CSA (MID-FC/csa_models.py:211-212, 218-219, 232-240):
//   pooled descriptors  y_k = mean_n SSA(x_k)                    -> csn_rowsum (fp64 accumulation)
//   mixed features      out = sum_k comp_k * LayerNormAffine(xhat_k)   -> csn_mix_fwd / csn_mix_bwd
// These are HBM-bound streaming passes over channel-major rows ([channel][point], 16-byte lanes).
// The reductions (row sums, <dOut, xhat_k>) feed gradients that are differences of large numbers
// (d comp is ~1e-7 of its terms), so they accumulate in double: 64-bit adds are free next to HBM.
#include "csn_common.h"
#include "csn_kernels.h"

namespace {

__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

using namespace csn_mode;

// 4 consecutive values of a normalised map: fp32, or (X16: 16-bit activation maps) fp16
template <bool X16> CSN_DEVINL f32x4 load_x4(const float* __restrict__ row, int i) {
  if constexpr (X16) return from16x4<true>(*reinterpret_cast<const s16x4*>(reinterpret_cast<const short*>(row) + i));
  else return *reinterpret_cast<const f32x4*>(row + i);
}
// the row that starts `el` elements into a map tensor
template <bool X16> CSN_DEVINL const float* row_at(const float* base, long long el) {
  if constexpr (X16) return reinterpret_cast<const float*>(reinterpret_cast<const short*>(base) + el);
  else return base + el;
}

// out[r] = sum_n x[r*ld + n], n < n  (one work-group per row)
template <bool X16>
__global__ __launch_bounds__(256) void csn_rowsum_kernel(const float* __restrict__ x, float* __restrict__ out, int n,
                                                         long long ld) {
  __shared__ double red[4];
  const float* __restrict__ p = row_at<X16>(x, (long long)blockIdx.x * ld);
  double s = 0.0;
  for (int i = threadIdx.x * 4; i < n; i += 1024) {
    const f32x4 v = load_x4<X16>(p, i);
    s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[blockIdx.x] = (float)s;
}

// out[e][c] = sum_t ws[e][t][c]  (per-tile partial sums of the out-projection epilogue; fp64 accumulation, fixed order)
__global__ __launch_bounds__(256) void csn_partial_sums_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                               int tiles, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float* __restrict__ p = ws + (long long)blockIdx.y * tiles * C + c;
  double s = 0.0;
  for (int t = 0; t < tiles; ++t) s += (double)p[(long long)t * C];
  out[(long long)blockIdx.y * C + c] = (float)s;
}

// feats[b][c][n] = gamma[c] * sum_k comp[b][k] xhat[(b*K1+k)][c][n] + beta[c] * sum_k comp[b][k]
// xhat0 != null: the k = 0 maps live in their own tensor xhat0[b][c][n] and xhat holds the K1 - 1 others, [b*(K1-1) + k-1]
template <bool X16>
__global__ __launch_bounds__(256) void csn_mix_fwd_kernel(const float* __restrict__ xhat, const float* __restrict__ comp,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ feats, int K1, int C, int NP,
                                                          const float* __restrict__ xhat0) {
  const int c = blockIdx.x % C, b = blockIdx.x / C;
  float w[8];
  float csum = 0.f;
  for (int k = 0; k < K1; ++k) { w[k] = comp[b * K1 + k]; csum += w[k]; }
  const float g = gamma[c], bb = beta[c] * csum;
  const float* xk[8];
  for (int k = 0; k < K1; ++k)
    xk[k] = !xhat0 ? row_at<X16>(xhat, ((long long)(b * K1 + k) * C + c) * NP)
                   : (k == 0 ? row_at<X16>(xhat0, ((long long)b * C + c) * NP)
                             : row_at<X16>(xhat, ((long long)(b * (K1 - 1) + k - 1) * C + c) * NP));
  float* __restrict__ o = feats + ((long long)b * C + c) * NP;
  for (int i = threadIdx.x * 4; i < NP; i += 1024) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K1; ++k) {
      const f32x4 v = load_x4<X16>(xk[k], i);
      acc += v * w[k];
    }
    *reinterpret_cast<f32x4*>(o + i) = acc * g + bb;
  }
}

// dxhat[(b*K1+k)][c][n] = comp[b][k] gamma[c] dfeats[b][c][n]  (+ pool_grad[(b*K1+k)][c] if given, k = 0 only)
// rowdot[b][k][c] = sum_n dfeats[b][c][n] xhat[(b*K1+k)][c][n],   rowsum[b][c] = sum_n dfeats[b][c][n]
// X16: xhat / xhat0 are fp16 maps (16-bit activation maps; reductions only — no gradient maps are written in that form)
template <bool X16>
__global__ __launch_bounds__(256) void csn_mix_bwd_kernel(const float* __restrict__ dfeats, const float* __restrict__ xhat,
                                                          const float* __restrict__ comp, const float* __restrict__ gamma,
                                                          float* __restrict__ dxhat, float* __restrict__ rowdot,
                                                          float* __restrict__ rowsum, int K1, int C, int NP,
                                                          const float* __restrict__ xhat0, float* __restrict__ dxhat0) {
  __shared__ double red[4];
  const int c = blockIdx.x % C, b = blockIdx.x / C;
  float w[8];
  for (int k = 0; k < K1; ++k) w[k] = comp[b * K1 + k] * gamma[c];
  const float* __restrict__ d = dfeats + ((long long)b * C + c) * NP;
  const float* xk[8];
  float* ok[8];
  for (int k = 0; k < K1; ++k) {
    const long long off = !xhat0 ? ((long long)(b * K1 + k) * C + c) * NP
                                 : (k == 0 ? ((long long)b * C + c) * NP : ((long long)(b * (K1 - 1) + k - 1) * C + c) * NP);
    xk[k] = row_at<X16>((xhat0 && k == 0) ? xhat0 : xhat, off);
    float* const o = (xhat0 && k == 0) ? dxhat0 : dxhat;
    ok[k] = o ? o + off : nullptr;
  }
  const bool store = !X16 && (xhat0 ? dxhat0 != nullptr : dxhat != nullptr);    // no gradient maps wanted: only the reductions
  double dot[8];
  for (int k = 0; k < 8; ++k) dot[k] = 0.0;
  double sum = 0.0;
  for (int i = threadIdx.x * 4; i < NP; i += 1024) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(d + i);
    sum += ((double)g.x + (double)g.y) + ((double)g.z + (double)g.w);
    for (int k = 0; k < K1; ++k) {
      const f32x4 v = load_x4<X16>(xk[k], i);
      dot[k] += ((double)g.x * v.x + (double)g.y * v.y) + ((double)g.z * v.z + (double)g.w * v.w);
      if (store) *reinterpret_cast<f32x4*>(ok[k] + i) = g * w[k];
    }
  }
  for (int k = 0; k < K1; ++k) {
    const double t = block_sum(dot[k], red);
    if (threadIdx.x == 0) rowdot[((long long)b * K1 + k) * C + c] = (float)t;
  }
  const double t = block_sum(sum, red);
  if (threadIdx.x == 0) rowsum[(long long)b * C + c] = (float)t;
}

}  // namespace

int csn_launch_rowsum_f32(const float* x, float* out, long long rows, int n, long long ld, hipStream_t st, int x16) {
  if (rows <= 0) return 0;
  if (x16) hipLaunchKernelGGL(csn_rowsum_kernel<true>, dim3((unsigned)rows), dim3(256), 0, st, x, out, n, ld);
  else hipLaunchKernelGGL(csn_rowsum_kernel<false>, dim3((unsigned)rows), dim3(256), 0, st, x, out, n, ld);
  return (int)hipGetLastError();
}

int csn_launch_partial_sums_f32(const float* ws, float* out, long long rows_outer, int tiles, int C, hipStream_t st) {
  if (rows_outer <= 0) return 0;
  hipLaunchKernelGGL(csn_partial_sums_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)rows_outer), dim3(256), 0, st, ws, out,
                     tiles, C);
  return (int)hipGetLastError();
}

int csn_launch_mix_fwd_f32(const float* xhat, const float* comp, const float* gamma, const float* beta, float* feats, int B,
                           int K1, int C, int NP, const float* xhat0, hipStream_t st, int x16) {
  if (x16) hipLaunchKernelGGL(csn_mix_fwd_kernel<true>, dim3((unsigned)(B * C)), dim3(256), 0, st, xhat, comp, gamma, beta, feats, K1, C,
                              NP, xhat0);
  else hipLaunchKernelGGL(csn_mix_fwd_kernel<false>, dim3((unsigned)(B * C)), dim3(256), 0, st, xhat, comp, gamma, beta, feats, K1, C,
                          NP, xhat0);
  return (int)hipGetLastError();
}

int csn_launch_mix_bwd_f32(const float* dfeats, const float* xhat, const float* comp, const float* gamma, float* dxhat,
                           float* rowdot, float* rowsum, int B, int K1, int C, int NP, const float* xhat0, float* dxhat0,
                           hipStream_t st, int x16) {
  if (x16) {
    if (dxhat || dxhat0) return -1;                                   // 16-bit maps: the linked form (reductions only)
    hipLaunchKernelGGL(csn_mix_bwd_kernel<true>, dim3((unsigned)(B * C)), dim3(256), 0, st, dfeats, xhat, comp, gamma, dxhat, rowdot,
                       rowsum, K1, C, NP, xhat0, dxhat0);
  } else
    hipLaunchKernelGGL(csn_mix_bwd_kernel<false>, dim3((unsigned)(B * C)), dim3(256), 0, st, dfeats, xhat, comp, gamma, dxhat, rowdot,
                       rowsum, K1, C, NP, xhat0, dxhat0);
  return (int)hipGetLastError();
}
