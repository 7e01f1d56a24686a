// Masked cross-entropy on class-major logits — the loss the reference trains the CSA layers with (MID-FC/csa_training.py:94-108:
// transpose the logits to [point][class], gather the points with label > mask, F.cross_entropy + accuracy over those) — as one
// pass over the logits where they lie, [shape][class][point], and one pass for the gradient:
//   forward   per point: lse = log sum_c exp(z_c), loss = lse - z_label, hit = (argmax_c z_c == label); per-block partial sums
//             of (loss, counted points, hits) in fp64, added in a fixed order by a second small kernel
//   backward  dz[s][c][n] = counted ? (exp(z_c - lse) - [c == label]) * g / count : 0
// Both are byte streams: 4 n_classes bytes per point forward, 8 n_classes backward (39 classes, 320 000 points: 50 / 100 MB).
#include "csn_common.h"
#include "csn_kernels.h"

namespace {

constexpr int CE_BLOCK = 256;

// thread = point; the classes of a point are n_points apart, so a wave reads 256 contiguous bytes per class
__global__ __launch_bounds__(CE_BLOCK) void csn_masked_ce_fwd_kernel(CsnMaskedCeArgs p) {
  __shared__ double red[3][CE_BLOCK / 64];
  const int n = blockIdx.x * CE_BLOCK + threadIdx.x, s = blockIdx.y;
  double loss = 0.0, cnt = 0.0, hit = 0.0;
  if (n < p.n_points) {
    const long long lab = p.labels[(long long)s * p.label_shape_stride + n];
    const float* z = p.logits + (long long)s * p.shape_stride + n;
    float m = -INFINITY, sum = 0.f, zl = 0.f;
    int arg = 0;
    for (int c = 0; c < p.n_classes; ++c) {
      const float v = z[(long long)c * p.ld];
      if (v > m) {                                     // (strictly greater: the first maximum, like torch.argmax)
        sum = sum * __expf(m - v) + 1.f;
        m = v;
        arg = c;
      } else {
        sum += v == m ? 1.f : __expf(v - m);           // (v == m covers a logit of -inf while the maximum still is: -inf - -inf)
      }
      if (c == lab) zl = v;
    }
    const float lse = m + __logf(sum);
    p.lse[(long long)s * p.n_points + n] = lse;
    if (lab > p.mask && lab < p.n_classes) {
      loss = (double)(lse - zl);
      cnt = 1.0;
      hit = arg == lab ? 1.0 : 0.0;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    loss += __shfl_xor(loss, off, 64);
    cnt += __shfl_xor(cnt, off, 64);
    hit += __shfl_xor(hit, off, 64);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = loss; red[1][wave] = cnt; red[2][wave] = hit; }
  __syncthreads();
  if (threadIdx.x < 3) {
    double t = 0.0;
    for (int w = 0; w < CE_BLOCK / 64; ++w) t += red[threadIdx.x][w];
    p.partials[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = t;
  }
}

// out[0] = mean loss over the counted points (0 / 0 = nan when there are none, like F.cross_entropy of an empty selection),
// out[1] = accuracy, out[2] = counted points
__global__ __launch_bounds__(256) void csn_masked_ce_finish_kernel(const double* __restrict__ partials, long long n_blocks, float* __restrict__ out) {
  __shared__ double red[3][256];
  double a[3] = {0.0, 0.0, 0.0};
  for (long long i = threadIdx.x; i < n_blocks; i += 256)
    for (int k = 0; k < 3; ++k) a[k] += partials[i * 3 + k];
  for (int k = 0; k < 3; ++k) red[k][threadIdx.x] = a[k];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st)
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = (float)(red[0][0] / red[1][0]);
    out[1] = (float)(red[2][0] / red[1][0]);
    out[2] = (float)red[1][0];
  }
}

// thread = 4 points of one class row (16-byte accesses), or — VEC = false — one point (any point count, pitch and alignment:
// the reference's loss takes any N, csa_training.py:94-108); grid (point groups, classes, shapes)
template <bool VEC>
__global__ __launch_bounds__(CE_BLOCK) void csn_masked_ce_bwd_kernel(CsnMaskedCeArgs p) {
  constexpr int W = VEC ? 4 : 1;
  const int n = (blockIdx.x * CE_BLOCK + threadIdx.x) * W, c = blockIdx.y, s = blockIdx.z;
  if (n >= p.n_points) return;
  const float scale = p.grad_out[0] / p.stats[2];
  const long long row = (long long)s * p.shape_stride + (long long)c * p.ld + n;
  const long long dro = (long long)s * p.dshape_stride + (long long)c * p.dld + n;
  const long long* lp = p.labels + (long long)s * p.label_shape_stride + n;
  float z[W], lse[W], d[W];
  if constexpr (VEC) {
    const f32x4 zv = *reinterpret_cast<const f32x4*>(p.logits + row);
    const f32x4 lv = *reinterpret_cast<const f32x4*>(p.lse + (long long)s * p.n_points + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) { z[i] = zv[i]; lse[i] = lv[i]; }
  } else {
    z[0] = p.logits[row];
    lse[0] = p.lse[(long long)s * p.n_points + n];
  }
#pragma unroll
  for (int i = 0; i < W; ++i) {
    const long long lab = lp[i];
    const bool counted = lab > p.mask && lab < p.n_classes;
    d[i] = counted ? (__expf(z[i] - lse[i]) - (lab == c ? 1.f : 0.f)) * scale : 0.f;
  }
  if constexpr (VEC) *reinterpret_cast<f32x4*>(p.dlogits + dro) = f32x4{d[0], d[1], d[2], d[3]};
  else p.dlogits[dro] = d[0];
}

}  // namespace

long long csn_masked_ce_blocks(int n_shapes, int n_points) { return (long long)n_shapes * ((n_points + CE_BLOCK - 1) / CE_BLOCK); }

int csn_launch_masked_ce_fwd(const CsnMaskedCeArgs& a, hipStream_t st) {
  const int bx = (a.n_points + CE_BLOCK - 1) / CE_BLOCK;
  hipLaunchKernelGGL(csn_masked_ce_fwd_kernel, dim3(bx, a.n_shapes), dim3(CE_BLOCK), 0, st, a);
  hipLaunchKernelGGL(csn_masked_ce_finish_kernel, dim3(1), dim3(256), 0, st, a.partials, (long long)bx * a.n_shapes, a.stats);
  return (int)hipGetLastError();
}

int csn_launch_masked_ce_bwd(const CsnMaskedCeArgs& a, hipStream_t st) {
  // 16-byte accesses where every row of logits, lse and dlogits starts 16-byte aligned and holds whole groups of 4 points
  const bool vec = !(a.n_points & 3) && !(a.ld & 3) && !(a.dld & 3) && !(a.shape_stride & 3) && !(a.dshape_stride & 3) &&
                   !(reinterpret_cast<uintptr_t>(a.logits) & 15) && !(reinterpret_cast<uintptr_t>(a.dlogits) & 15) &&
                   !(reinterpret_cast<uintptr_t>(a.lse) & 15);
  if (vec) {
    const int bx = (a.n_points / 4 + CE_BLOCK - 1) / CE_BLOCK;
    hipLaunchKernelGGL(csn_masked_ce_bwd_kernel<true>, dim3(bx, a.n_classes, a.n_shapes), dim3(CE_BLOCK), 0, st, a);
  } else {
    const int bx = (a.n_points + CE_BLOCK - 1) / CE_BLOCK;
    hipLaunchKernelGGL(csn_masked_ce_bwd_kernel<false>, dim3(bx, a.n_classes, a.n_shapes), dim3(CE_BLOCK), 0, st, a);
  }
  return (int)hipGetLastError();
}
