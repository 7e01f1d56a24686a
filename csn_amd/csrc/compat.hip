// The compatibility head of the cross-shape attention layer (MID-FC/csa_models.py:222-230) and its autograd, fused:
//   u_q = normalize(W_q y_0 + b_q),  u_k = normalize(W_k y_k + b_k),  comp = softmax_k <u_q, u_k>      per query shape
// y = the pooled SSA descriptors (B, K+1, C) after the LayerNorm affine.  In torch this is two nn.Linear calls on (32, 256) and
// (128, 256) rows, two F.normalize, an einsum and a softmax — nine library GEMM launches of 15-55 us each plus ~35 elementwise
// and reduction launches per training step (0.35 ms of a 27.7 ms step) for 63 MFLOP.  Here: one launch forward (a work-group per
// query shape, a thread per output channel), two launches backward (per-shape vectors and input gradients; then the weight /
// bias gradients as sums over the shapes in a FIXED order: bitwise reproducible).  All sums accumulate in fp64: on inputs
// whose descriptors nearly coincide the head's gradients are 1e-7-sized differences of O(1) sums (golden set G4),
// and 63 MFLOP of fp64 cost nothing next to the launches they replace.
// `reference_layout`: the reference concatenates the key descriptors neighbour-major and re-views them as (B, K+1, C)
// (csa_models.py:213,220,227), so for B > 1 the keys of shape b are the rows b (K+1) .. of that stack: key row (b, k) is the
// descriptor of shape (b (K+1) + k) % B, neighbour slot (b (K+1) + k) / B.  0 = every shape against its own neighbours.
#include "csn_common.h"
#include "csn_kernels.h"

namespace {

constexpr int CMAX = 256, K1MAX = 8, UNR = 16;

// key row (b, k) of the head -> row of the (B, K1, C) descriptor tensor
CSN_DEVINL int key_src_row(int b, int k, int B, int K1, int reference_layout) {
  if (!reference_layout) return b * K1 + k;
  const int r = b * K1 + k;                       // row of the neighbour-major stack: (slot r / B, shape r % B)
  return (r % B) * K1 + r / B;
}

// sum over the work-group's threads (256): every thread gets the total
CSN_DEVINL double block_sum(double v, double* red, int tid) {
  v += __shfl_xor(v, 32, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 4, 64);  v += __shfl_xor(v, 2, 64);  v += __shfl_xor(v, 1, 64);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// forward.  wq_t / wk_t: the weights TRANSPOSED, [in][out] (consecutive threads = consecutive outputs read consecutive words)
template <int K1>
__global__ __launch_bounds__(256) void csn_compat_fwd_kernel(const float* __restrict__ pooled, const float* __restrict__ wq_t,
                                                             const float* __restrict__ bq, const float* __restrict__ wk_t,
                                                             const float* __restrict__ bk, float* __restrict__ comp,
                                                             double* __restrict__ save_u, double* __restrict__ save_n, int B, int C,
                                                             int reference_layout) {
  __shared__ float x[1 + K1][CMAX];
  __shared__ double red[4];
  __shared__ double sc[K1];
  const int b = blockIdx.x, t = threadIdx.x;
  const bool on = t < C;
  for (int v = 0; v <= K1; ++v) {
    const int row = v == 0 ? b * K1 : key_src_row(b, v - 1, B, K1, reference_layout);
    x[v][t] = on ? pooled[(long long)row * C + t] : 0.f;
  }
  __syncthreads();
  double raw[1 + K1];
  raw[0] = on ? (double)bq[t] : 0.0;
  for (int v = 1; v <= K1; ++v) raw[v] = on ? (double)bk[t] : 0.0;
  if (on) {
    // the weight words of UNR inputs are requested together: the loop is bound by the latency of these loads, nothing else
    int j0 = 0;
    for (; j0 + UNR <= C; j0 += UNR) {
      float a[UNR], c[UNR];
#pragma unroll
      for (int jj = 0; jj < UNR; ++jj) {
        a[jj] = wq_t[(long long)(j0 + jj) * C + t];
        c[jj] = wk_t[(long long)(j0 + jj) * C + t];
      }
#pragma unroll
      for (int jj = 0; jj < UNR; ++jj) {
        raw[0] = fma((double)a[jj], (double)x[0][j0 + jj], raw[0]);
#pragma unroll
        for (int v = 1; v <= K1; ++v)
          raw[v] = fma((double)c[jj], (double)x[v][j0 + jj], raw[v]);
      }
    }
    for (; j0 < C; ++j0) {
      const double a = wq_t[(long long)j0 * C + t], c = wk_t[(long long)j0 * C + t];
      raw[0] = fma(a, (double)x[0][j0], raw[0]);
#pragma unroll
      for (int v = 1; v <= K1; ++v)
        raw[v] = fma(c, (double)x[v][j0], raw[v]);
    }
  }
  double u[1 + K1];
#pragma unroll
  for (int v = 0; v <= K1; ++v)
    {
      const double n = fmax(sqrt(block_sum(raw[v] * raw[v], red, t)), 1e-12);          // F.normalize: x / max(|x|, eps)
      u[v] = raw[v] / n;
      if (on) save_u[((long long)b * (K1 + 1) + v) * C + t] = u[v];
      if (t == 0) save_n[b * (K1 + 1) + v] = n;
    }
#pragma unroll
  for (int v = 1; v <= K1; ++v)
    {
      const double s = block_sum(u[0] * u[v], red, t);
      if (t == 0) sc[v - 1] = s;
    }
  __syncthreads();
  if (t == 0) {
    double m = sc[0];
    for (int k = 1; k < K1; ++k) m = fmax(m, sc[k]);
    double e[K1], z = 0.0;
    for (int k = 0; k < K1; ++k) { e[k] = exp(sc[k] - m); z += e[k]; }
    for (int k = 0; k < K1; ++k) comp[b * K1 + k] = (float)(e[k] / z);
  }
}

// backward, per query shape: softmax and normalisation backward, d_raw (for the weight gradients) and the input gradients
// dx = W^T d_raw in the head's row space (row 0: the query descriptor, rows 1..K1: the key rows)
template <int K1>
__global__ __launch_bounds__(256) void csn_compat_bwd_rows_kernel(const float* __restrict__ dcomp, const float* __restrict__ comp,
                                                                  const double* __restrict__ save_u, const double* __restrict__ save_n,
                                                                  const float* __restrict__ wq, const float* __restrict__ wk,
                                                                  double* __restrict__ d_raw, double* __restrict__ dx, int B, int C) {
  __shared__ double dr[1 + K1][CMAX];
  __shared__ double red[4];
  const int b = blockIdx.x, t = threadIdx.x;
  const bool on = t < C;
  double dsk[K1], dot = 0.0;
  for (int k = 0; k < K1; ++k) dot += (double)comp[b * K1 + k] * (double)dcomp[b * K1 + k];
  for (int k = 0; k < K1; ++k) dsk[k] = (double)comp[b * K1 + k] * ((double)dcomp[b * K1 + k] - dot);   // d softmax
  double u[1 + K1], du[1 + K1];
  for (int v = 0; v <= K1; ++v) u[v] = on ? save_u[((long long)b * (K1 + 1) + v) * C + t] : 0.0;
  du[0] = 0.0;
  for (int k = 0; k < K1; ++k) { du[0] = fma(dsk[k], u[k + 1], du[0]); du[k + 1] = dsk[k] * u[0]; }
#pragma unroll
  for (int v = 0; v <= K1; ++v)
    {
      const double n = save_n[b * (K1 + 1) + v];
      const double ud = block_sum(u[v] * du[v], red, t);
      const double g = n > 1e-12 ? (du[v] - u[v] * ud) / n : du[v] / 1e-12;             // (below eps the division is by a constant)
      dr[v][t] = on ? g : 0.0;
      if (on) d_raw[((long long)b * (K1 + 1) + v) * C + t] = g;
    }
  __syncthreads();
  if (on) {
    double acc[1 + K1];
    for (int v = 0; v <= K1; ++v) acc[v] = 0.0;
    int i0 = 0;
    for (; i0 + UNR <= C; i0 += UNR) {
      float a[UNR], c[UNR];
#pragma unroll
      for (int ii = 0; ii < UNR; ++ii) {
        a[ii] = wq[(long long)(i0 + ii) * C + t];
        c[ii] = wk[(long long)(i0 + ii) * C + t];
      }
#pragma unroll
      for (int ii = 0; ii < UNR; ++ii) {
        acc[0] = fma((double)a[ii], dr[0][i0 + ii], acc[0]);
#pragma unroll
        for (int v = 1; v <= K1; ++v)
          acc[v] = fma((double)c[ii], dr[v][i0 + ii], acc[v]);
      }
    }
    for (; i0 < C; ++i0) {
      const double a = wq[(long long)i0 * C + t], c = wk[(long long)i0 * C + t];
      acc[0] = fma(a, dr[0][i0], acc[0]);
#pragma unroll
      for (int v = 1; v <= K1; ++v)
        acc[v] = fma(c, dr[v][i0], acc[v]);
    }
    for (int v = 0; v <= K1; ++v) dx[((long long)b * (K1 + 1) + v) * C + t] = acc[v];
  }
}

// backward, sums over the shapes (fixed order).  blockIdx.y: 0 dW_q, 1 dW_k (one thread per weight), 2: biases and the
// gather of the input gradients back into descriptor rows
__global__ __launch_bounds__(256) void csn_compat_bwd_sums_kernel(const float* __restrict__ pooled, const double* __restrict__ d_raw,
                                                                  const double* __restrict__ dx, float* __restrict__ dwq,
                                                                  float* __restrict__ dbq, float* __restrict__ dwk,
                                                                  float* __restrict__ dbk, float* __restrict__ dpooled, int B, int K1,
                                                                  int C, int reference_layout) {
  const int what = blockIdx.y;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (what < 2) {
    if (idx >= (long long)C * C) return;
    const int i = (int)(idx / C), j = (int)(idx % C);
    double s = 0.0;
    if (what == 0) {
#pragma unroll 8
      for (int b = 0; b < B; ++b) s = fma(d_raw[((long long)b * (K1 + 1)) * C + i], (double)pooled[((long long)b * K1) * C + j], s);
      dwq[idx] = (float)s;
    } else {
#pragma unroll 8
      for (int r = 0; r < B * K1; ++r) {                              // r = b K1 + k
        const int b = r / K1;
        s = fma(d_raw[((long long)r + b + 1) * C + i], (double)pooled[(long long)key_src_row(b, r - b * K1, B, K1, reference_layout) * C + j], s);
      }
      dwk[idx] = (float)s;
    }
    return;
  }
  if (idx < C) {                                                     // biases
    double sq = 0.0, sk = 0.0;
    for (int b = 0; b < B; ++b) {
      sq += d_raw[((long long)b * (K1 + 1)) * C + idx];
      for (int k = 0; k < K1; ++k) sk += d_raw[((long long)b * (K1 + 1) + 1 + k) * C + idx];
    }
    dbq[idx] = (float)sq;
    dbk[idx] = (float)sk;
  }
  // descriptor row (b, k): its key row is the inverse of key_src_row; slot 0 also carries the query-side gradient
  if (idx < (long long)B * K1 * C) {
    const int row = (int)(idx / C), c = (int)(idx % C);
    const int b = row / K1, k = row % K1;
    int hb = b, hk = k;
    if (reference_layout) { const int r = k * B + b; hb = r / K1; hk = r % K1; }
    double g = dx[((long long)hb * (K1 + 1) + 1 + hk) * C + c];
    if (k == 0) g += dx[((long long)b * (K1 + 1)) * C + c];
    dpooled[idx] = (float)g;
  }
}

}  // namespace

int csn_launch_compat_fwd(const float* pooled, const float* wq_t, const float* bq, const float* wk_t, const float* bk, float* comp,
                          double* save_u, double* save_n, int B, int K1, int C, int reference_layout, hipStream_t st) {
  if (K1 < 1 || K1 > K1MAX || C > CMAX) return -5;
#define CSN_COMPAT_FWD(K)                                                                                                         \
  case K:                                                                                                                         \
    hipLaunchKernelGGL(csn_compat_fwd_kernel<K>, dim3(B), dim3(256), 0, st, pooled, wq_t, bq, wk_t, bk, comp, save_u, save_n, B, C, \
                       reference_layout);                                                                                         \
    break;
  switch (K1) {
    CSN_COMPAT_FWD(1) CSN_COMPAT_FWD(2) CSN_COMPAT_FWD(3) CSN_COMPAT_FWD(4) CSN_COMPAT_FWD(5) CSN_COMPAT_FWD(6) CSN_COMPAT_FWD(7)
    CSN_COMPAT_FWD(8)
  }
#undef CSN_COMPAT_FWD
  return (int)hipGetLastError();
}

int csn_launch_compat_bwd(const float* dcomp, const float* comp, const double* save_u, const double* save_n, const float* pooled,
                          const float* wq, const float* wk, double* d_raw, double* dx, float* dpooled, float* dwq, float* dbq,
                          float* dwk, float* dbk, int B, int K1, int C, int reference_layout, hipStream_t st) {
  if (K1 < 1 || K1 > K1MAX || C > CMAX) return -5;
#define CSN_COMPAT_BWD(K)                                                                                                          \
  case K:                                                                                                                          \
    hipLaunchKernelGGL(csn_compat_bwd_rows_kernel<K>, dim3(B), dim3(256), 0, st, dcomp, comp, save_u, save_n, wq, wk, d_raw, dx, B, C); \
    break;
  switch (K1) {
    CSN_COMPAT_BWD(1) CSN_COMPAT_BWD(2) CSN_COMPAT_BWD(3) CSN_COMPAT_BWD(4) CSN_COMPAT_BWD(5) CSN_COMPAT_BWD(6) CSN_COMPAT_BWD(7)
    CSN_COMPAT_BWD(8)
  }
#undef CSN_COMPAT_BWD
  const long long n = (long long)C * C > (long long)B * K1 * C ? (long long)C * C : (long long)B * K1 * C;
  hipLaunchKernelGGL(csn_compat_bwd_sums_kernel, dim3((unsigned)((n + 255) / 256), 3), dim3(256), 0, st, pooled, d_raw, dx, dwq, dbq,
                     dwk, dbk, dpooled, B, K1, C, reference_layout);
  return (int)hipGetLastError();
}
