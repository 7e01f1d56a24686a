// Weight gradients of the K = 256 products in the bf16x3 mode as a STREAM over the points (autograd of MID-FC/csa_models.py:
// 103-105 and :115 w.r.t. their weights):   dW[r][c] = sum_{item, n} A[item][r][n] B[item][c][n]
// (dW_fc: A = dz, B = Ctx; dW_q | dW_k | dW_v: A = the stacked dQ / dK / dV maps, B = x).  Both operands are read ONCE, nothing
// is written inside the loop: 2 KB per point and 256-row set in, 131 kFLOP — the same balance as the products of wx_stream.hip,
// so the floor is the read stream.  The tiled kernel (gemm_bf16x3.hip, 256 x 256 output tiles over slabs of ~5000 points, then a
// slab reduction) ran them at 3.9-4.8 TB/s with the matrix pipe 49 % busy.
// Here the OUTPUT is stationary: a persistent work-group keeps one 256 x 256 block of dW in its accumulators (wave w: rows
// 32 w .. + 31 as 8 tiles of v_mfma_f32_32x32x16_bf16 = 128 registers) and streams 32-point chunks of A and of B (32 KB each)
// through two LDS stages of hi / lo planes; a chunk is 48 matrix instructions per wave (K = 32 points: 2 steps x 8 column
// tiles x 3 products).  Both operands are read from LDS as plain 16-byte fragments (8 consecutive points of a row); units are
// XOR-swizzled by (row / 4) % 4 so that the 16 rows of a read pass cover every bank once.  One barrier per chunk; the commit of
// chunk c + 1 (split + LDS stores) and the contraction of chunk c touch different stages.  Two chunks in flight in registers.
// At the end every work-group writes its block as one slab of the workspace and csn_slab_reduce_kernel adds the slabs in a fixed
// order (bitwise reproducible), applying scale / accumulate.
#include "csn_common.h"
#include "csn_kernels.h"
#include "wx_common.h"
#include <type_traits>

namespace {

constexpr int WG_OP = 2 * WX_PLANE;             // 16-bit elements of one operand's hi + lo planes of a stage (32 KB)
constexpr int WG_STAGE = 2 * WG_OP;             // A and B

template <int N>
CSN_DEVINL void wg_arrived(f32x4* R) {
  asm volatile("s_waitcnt vmcnt(%8)" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]), "+v"(R[5]), "+v"(R[6]), "+v"(R[7]) : "n"(N) : "memory");
}

__global__ __launch_bounds__(512, 2) void csn_wx_wgrad_kernel(CsnWxWgradArgs p) {
  __shared__ __attribute__((aligned(16))) short smem[2 * WG_STAGE];            // 2 stages x [A hi | A lo | B hi | B lo] = 128 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;

  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int spx = (int)(gridDim.x >> 3) / p.n_sets;
  if (j >= spx * p.n_sets) return;
  const int set = j % p.n_sets, n_streams = spx * 8;
  const int stream = (j / p.n_sets) * 8 + xcd;
  const unsigned cpi = (unsigned)(p.n_points + WX_CH - 1) / WX_CH;
  const int n_chunks = p.n_items * (int)cpi;

  // staging: thread -> rows tid / 8 + 64 i of both operands, points 4 (tid % 8) .. + 3
  const int krow = tid >> 3, c4 = tid & 7;
  const unsigned a_voff = (unsigned)(krow * p.lda + 4 * c4) * 4u, b_voff = (unsigned)(krow * p.ldb + 4 * c4) * 4u;
  auto issue = [&](int q, f32x4* R) {
    const bool exists = q < n_chunks && !(p.ablate & 4);
    const unsigned item = __builtin_amdgcn_readfirstlane(exists ? (unsigned)q / cpi : 0u);
    const int col0 = exists ? (int)(((unsigned)q - item * cpi) * WX_CH) : 0;
    const int valid = exists ? min(WX_CH, p.n_points - col0) : 0;
    const u32x4 Ar = wx_rsrc(p.a + (long long)item * p.a_stride + (long long)(256 * set) * p.lda + col0, ((long long)255 * p.lda + valid) * 4);
    const u32x4 Br = wx_rsrc(p.b + (long long)item * p.b_stride + col0, ((long long)255 * p.ldb + valid) * 4);
    const bool on = 4 * c4 < valid;
#pragma unroll
    for (int i = 0; i < 4; ++i) wx_request(R[i], Ar, on ? a_voff : CSN_OOB, (unsigned)(64 * i * p.lda) * 4u);
#pragma unroll
    for (int i = 0; i < 4; ++i) wx_request(R[4 + i], Br, on ? b_voff : CSN_OOB, (unsigned)(64 * i * p.ldb) * 4u);
  };
  // LDS image of an operand plane: [256 rows][32 points] bf16, 64 bytes a row; 16-byte unit u of row r sits at u ^ ((r >> 2) & 3)
  const int c_dst = krow * WX_CH + 8 * ((c4 >> 1) ^ ((krow >> 2) & 3)) + 4 * (c4 & 1);       // (rows + 64 i: the same swizzle)
  auto commit = [&](int stage, const f32x4* R) {
    short* dst = smem + stage * WG_STAGE + c_dst;
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s16x4 hi, lo;
        split4<Bf16x3>(R[4 * o + i], hi, lo);
        *reinterpret_cast<s16x4*>(dst + o * WG_OP + 64 * i * WX_CH) = hi;
        *reinterpret_cast<s16x4*>(dst + o * WG_OP + 64 * i * WX_CH + WX_PLANE) = lo;
      }
  };
  // fragments: lane l reads the 8 points 16 s + 8 h .. + 7 of row 32 t + (l & 31): unit 2 s + h, swizzled by the row
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int f_row = l31 * WX_CH, f_sw = (l31 >> 2) & 3;              // (rows 32 t + l31: (row >> 2) & 3 = (l31 >> 2) & 3)
  auto frag = [&](const short* plane, int row0, int s) {
    return *reinterpret_cast<const s16x8*>(plane + row0 * WX_CH + f_row + 8 * ((2 * s + h) ^ f_sw));
  };
  auto compute = [&](int stage) {
    if (p.ablate & 1) {
      acc[0][0] += __builtin_bit_cast(float, (int)smem[stage * WG_STAGE + tid]);
      return;
    }
    const short* A = smem + stage * WG_STAGE;
    const short* B = A + WG_OP;
    // the wave's A fragments of both steps, then the 16 (step, column tile) products with the B fragments read two ahead
    // (explicit ring, order pinned: left alone the compiler hoists all 32 fragment reads — 128 registers — and spills)
    s16x8 ah[2], al[2], bh[2], bl[2];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 2; ++s) { ah[s] = frag(A, 32 * wave, s); al[s] = frag(A + WX_PLANE, 32 * wave, s); }
#pragma unroll
    for (int i = 0; i < 2; ++i) { bh[i] = frag(B, 32 * (i & 7), i >> 3); bl[i] = frag(B + WX_PLANE, 32 * (i & 7), i >> 3); }
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = i & 7, s = i >> 3, r = i & 1;
      acc[t] = wx_mma(ah[s], al[s], bh[r], bl[r], acc[t]);
      if (i + 2 < 16) { bh[r] = frag(B, 32 * ((i + 2) & 7), (i + 2) >> 3); bl[r] = frag(B + WX_PLANE, 32 * ((i + 2) & 7), (i + 2) >> 3); }
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // chunk i of this stream is stream + n_streams i; chunk k travels in register set k % 2 and lands in stage k % 2
  f32x4 R0[8], R1[8];
  const int n_mine = stream < n_chunks ? (n_chunks - stream + n_streams - 1) / n_streams : 0;
  int q = stream;
  issue(q, R0); q += n_streams;
  issue(q, R1); q += n_streams;
  wg_arrived<8>(R0);
  commit(0, R0);
  __syncthreads();
  for (int c = 0; c < n_mine; c += 2) {
    issue(q, R0); q += n_streams;                 // chunk c + 2
    wg_arrived<8>(R1);                            // chunk c + 1: behind it only the request just made
    commit(1, R1);
    compute(0);
    __syncthreads();
    if (c + 1 >= n_mine) break;
    issue(q, R1); q += n_streams;                 // chunk c + 3
    wg_arrived<8>(R0);                            // chunk c + 2
    commit(0, R0);
    compute(1);
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // the block leaves as slab `stream` of the workspace: [slab][256 n_sets rows][256]
  float* out = p.ws + ((long long)stream * (256 * p.n_sets) + 256 * set + 32 * wave) * 256;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(long long)csn_acc_row(r, h) * 256 + 32 * t + l31] = acc[t][r];
}

}  // namespace

// the number of slabs a launch writes (0: the product is not taken); ws must hold slabs * rows * 256 floats
int csn_wx_wgrad_slabs(int rows, int cols) {
  // MEASURED AND NOT TAKEN (profiles/r4ae_weight_gradient_stream.txt): under the profiler dW_fc runs 1.06 ms here against 1.34 ms
  // on the tiled kernel (+ 0.07 of slab reduction), dW_q | dW_k | dW_v (three sets: the work-groups of a stream drift apart and
  // every set fetches the B chunks for itself) 1.75 against 1.34 — and the config-3 step does not move either way (26.10 / 26.17
  // against 26.08 ms).  Off by default: CSN_DEV_WX bit 8 switches it on for one 256-row set, bit 9 for up to four.
  if (!(csn_dev_wx & 1) || !(csn_dev_wx & 768) || cols != 256 || rows <= 0 || rows % 256 || rows / 256 > ((csn_dev_wx & 512) ? 4 : 1)) return 0;
  const int n_sets = rows / 256;
  return ((wx_grid() >> 3) / n_sets) * 8;
}

int csn_launch_wx_wgrad(const CsnWxWgradArgs& a, hipStream_t st) {
  if (a.n_items <= 0 || a.n_points <= 0) return -1;
  if ((a.lda & 3) || (a.ldb & 3) || (a.n_points & 3)) return -2;
  if ((long long)a.n_items * ((a.n_points + WX_CH - 1) / WX_CH + 16) * 2 + 8ll * wx_grid() >= (1ll << 31)) return CSN_NOT_TAKEN;
  CsnWxWgradArgs b = a;
  b.ablate = (csn_dev_wx >> 4) & 15;
  hipLaunchKernelGGL(csn_wx_wgrad_kernel, dim3(wx_grid()), dim3(512), 0, st, b);
  return (int)hipGetLastError();
}
