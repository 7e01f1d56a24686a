// Weight-stationary streaming products  Out[item][m][n] = sum_k W[m][k] X[item][k][n]  for K = 256 (d_model of the reference,
// MID-FC/csa_models.py:49-52, 147) in the bf16x3 math mode: the Q / K / V projections (csa_models.py:103-105), the gradient of
// the attention output dCtx = W_fc^T dZ (backward of csa_models.py:115) and the output projection with its residual + LayerNorm
// epilogue (csa_models.py:115-118).
//
// Why not the tiled GEMM of gemm_bf16x3.hip: with K = 256 these products move 64-128 FLOP per byte — below the bf16x3 ridge —
// so their floor is the byte stream, and a 256 x 256 tile kernel spends a quarter to a half of every tile in a prologue /
// epilogue with nothing in flight, and re-stages (and re-splits) the same 256 x 256 weight for every tile.  Here the weight
// never moves again after the first microsecond:
//   * persistent grid, one 8-wave work-group per CU; wave w keeps rows 32 w .. 32 w + 31 of W as bf16 hi / lo A fragments of
//     v_mfma_f32_32x32x16_bf16 in 128 registers for the whole launch;
//   * X streams through in chunks of 32 points (256 x 32 fp32 = 32 KB): two chunks in flight in registers, three stages of
//     bf16 hi / lo planes in LDS ([k][32 points], read back with the transposing ds_read_b64_tr_b16), ONE barrier per chunk;
//   * a chunk is 48 matrix instructions per wave (1.5 k cycles; 3 k per SIMD) against 6.5 k cycles of HBM time for its
//     64 KB in + out — every wave's epilogue (a private 4 KB LDS transpose, 16-byte row stores) runs beside other waves'
//     matrix work, and the load of chunk c + 3 is issued before chunk c is touched.
// Row sets: W may hold several 256-row sets (K and V: 2; H heads: 3 H).  Work-groups with equal blockIdx % 8 share an XCD's
// L2 (speed only); the sets of one stream sit on such work-groups and walk the same chunks, so X leaves HBM once.
#include "csn_common.h"
#include "csn_kernels.h"

namespace {

using namespace csn_mode;
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;

constexpr int WX_K = 256;                      // contraction length = rows per set
constexpr int WX_CH = 32;                      // points per chunk
constexpr int WX_NS = 3;                       // LDS stages
constexpr int WX_PLANE = WX_K * WX_CH;         // 16-bit elements per plane of a stage
constexpr int WX_STAGE = 2 * WX_PLANE;         // hi + lo
constexpr int WX_EB = 32 * 32;                 // floats of a wave's epilogue block

CSN_DEVINL f32x16 wx_mma(s16x8 ah, s16x8 al, s16x8 bh, s16x8 bl, f32x16 c) {
  c = mfma32<false>(al, bh, c);
  c = mfma32<false>(ah, bl, c);
  return mfma32<false>(ah, bh, c);
}

// OUT 0: fp32 map [item][rows][ldo];  OUT 2: bf16 tile planes (attn_bf16x3.hip): per row and block of tb points 16 tiles of
// [hi 32 | lo 32], block pitch 1024, row pitch ldo 16-bit elements, the padding keys of a block's last tile written as zeros
template <int OUT>
__global__ __launch_bounds__(512, 2) void csn_wx_kernel(CsnWxArgs p) {
  __shared__ __attribute__((aligned(16))) short smem[WX_NS * WX_STAGE + 8 * WX_EB * 2];      // 96 KB + 32 KB
  short* xs = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  float* eb = reinterpret_cast<float*>(smem + WX_NS * WX_STAGE) + wave * WX_EB;

  // streams: work-group b sits on XCD label b % 8; its place j = b / 8 there is (stream lane, row set)
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int spx = (int)(gridDim.x >> 3) / p.n_sets;
  if (j >= spx * p.n_sets) return;
  const int set = j % p.n_sets, n_streams = spx * 8;
  const int stream = (j / p.n_sets) * 8 + xcd;
  const unsigned cpi = (unsigned)(p.n_points + WX_CH - 1) / WX_CH;
  const int n_chunks = p.n_items * (int)cpi;                  // (the launcher keeps it below 2^31)
  if (stream >= n_chunks) return;
  // rows < div_rows are divided by div_val; a wave's 32 rows are all in or all out (div_rows % 32 == 0)
  const bool dv = 256 * set + 32 * wave < p.div_rows;
  const float dscale = (dv && p.div_exact) ? p.div_rcp : 1.f;
  const bool true_div = dv && !p.div_exact;

  // the wave's 32 rows of W as A fragments: lane l holds W[row l & 31][16 s + 8 (l >> 5) + j], j = 0..7
  s16x8 Wh[WX_K / 16], Wl[WX_K / 16];
  {
    const float* wrow = p.w + ((long long)(256 * set + 32 * wave + l31) * WX_K + 8 * h);
#pragma unroll
    for (int s = 0; s < WX_K / 16; ++s) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(wrow + 16 * s), a1 = *reinterpret_cast<const f32x4*>(wrow + 16 * s + 4);
      s16x4 h0, l0, h1, l1;
      split4<Bf16x3>(a0, h0, l0);
      split4<Bf16x3>(a1, h1, l1);
      Wh[s] = join8(h0, h1);
      Wl[s] = join8(l0, l1);
    }
  }

  // staging: a chunk is 256 k rows x 8 pieces of 16 bytes; thread -> k row tid / 8 + 64 i, piece tid % 8
  const int krow = tid >> 3, c4 = tid & 7;
  const unsigned x_voff = (unsigned)(krow * p.ldx + 4 * c4) * 4u;
  auto issue = [&](int q, f32x4* R) {
    if (q >= n_chunks) return;
    const unsigned item = (unsigned)q / cpi;
    const int col0 = (int)((unsigned)q - item * cpi) * WX_CH, valid = min(WX_CH, p.n_points - col0);
    const csn_rsrc_t Xr = csn_make_rsrc(p.x + (long long)item * p.x_item_stride + col0, ((long long)(WX_K - 1) * p.ldx + valid) * 4);
    const unsigned off = 4 * c4 < valid ? x_voff : CSN_OOB;
#pragma unroll
    for (int i = 0; i < 4; ++i) R[i] = csn_bload4(Xr, off, (unsigned)(64 * i * p.ldx) * 4u);
  };
  auto commit = [&](int q, int stage, const f32x4* R) {
    if (q >= n_chunks) return;
    short* dst = xs + stage * WX_STAGE + krow * WX_CH + 4 * c4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s16x4 hi, lo;
      split4<Bf16x3>(R[i], hi, lo);
      *reinterpret_cast<s16x4*>(dst + 64 * i * WX_CH) = hi;
      *reinterpret_cast<s16x4*>(dst + 64 * i * WX_CH + WX_PLANE) = lo;
    }
  };
  // B fragment of k step s: lane l holds X[16 s + 8 (l >> 5) + j][point l & 31].  Transposing read: 16-lane group g covers
  // points 16 (g & 1) .. + 15 and k rows 8 (g >> 1) .. + 7 in two passes of 4 rows; inside the group lane 4 q + p addresses
  // row q, points 4 p .. 4 p + 3.  A pass of a 32-lane half reads 4 consecutive 64-byte rows: every bank once.
  const int grp = lane >> 4, gq = (lane >> 2) & 3, gp = lane & 3;
  const int tr_base = (8 * (grp >> 1) + gq) * WX_CH + 16 * (grp & 1) + 4 * gp;
  auto compute = [&](int stage) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const short* xh = xs + stage * WX_STAGE + tr_base;
    // fragment reads run WX_PD k steps ahead of the matrix instructions that use them (register ring; the order is pinned)
    constexpr int WX_PD = 2, NSTEP = WX_K / 16;
    s16x8 bh[WX_PD], bl[WX_PD];
    auto rd = [&](int s, s16x8& fh, s16x8& fl) {
      const short* a = xh + 16 * s * WX_CH;
      fh = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a)),
                 __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + 4 * WX_CH)));
      fl = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + WX_PLANE)),
                 __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + WX_PLANE + 4 * WX_CH)));
    };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < WX_PD; ++s) rd(s, bh[s], bl[s]);
    __builtin_amdgcn_sched_group_barrier(0x100, 4 * WX_PD, 0);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      const int r = s % WX_PD;
      acc = wx_mma(Wh[s], Wl[s], bh[r], bl[r], acc);
      if (s + WX_PD < NSTEP) rd(s + WX_PD, bh[r], bl[r]);
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    return acc;
  };
  // epilogue: the wave's 32 x 32 block through its private LDS block (row-major, 128-byte rows) and out as 16-byte rows:
  // lane -> row lane / 8 + 8 t, points 4 (lane % 8) .. + 3
  const int erow = lane >> 3, c8 = lane & 7;
  auto epilogue = [&](int q, f32x16 acc) {
    const unsigned item = (unsigned)q / cpi;
    const int col0 = (int)((unsigned)q - item * cpi) * WX_CH;
    const int n = col0 + 4 * c8;
    const bool n_ok = n < p.n_points;
    if (true_div) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = acc[r] / p.div_val;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) eb[csn_acc_row(r, h) * 32 + l31] = acc[r] * dscale;
    f32x4 vals[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) vals[t] = *reinterpret_cast<const f32x4*>(&eb[(erow + 8 * t) * 32 + 4 * c8]);
    if constexpr (OUT == 0) {
      const csn_rsrc_t Or = csn_make_rsrc(reinterpret_cast<float*>(p.out) + (long long)item * p.out_item_stride + (long long)(256 * set) * p.ldo,
                                          (long long)256 * p.ldo * 4);
      const unsigned off = n_ok ? (unsigned)((32 * wave + erow) * p.ldo + n) * 4u : CSN_OOB;
#pragma unroll
      for (int t = 0; t < 4; ++t) csn_bstore4(vals[t], Or, off, (unsigned)(8 * t * p.ldo) * 4u);
    } else {
      const csn_rsrc_t Or = csn_make_rsrc(reinterpret_cast<short*>(p.out) + (long long)item * p.out_item_stride + (long long)(256 * set) * p.ldo,
                                          (long long)256 * p.ldo * 2);
      const int blk = n / p.tb, kib = n - blk * p.tb;                  // tb % 4 == 0: the 4 points share block and tile
      const unsigned tcol = (unsigned)(blk * 1024 + (kib >> 5) * 64 + (kib & 31));
      const unsigned off = n_ok ? ((unsigned)((32 * wave + erow) * p.ldo) + tcol) * 2u : CSN_OOB;
      int tpad = 0;                                                    // 4-key groups of zero padding behind this lane's points
      if (n_ok && (kib + 4 == p.tb || n + 4 == p.n_points)) tpad = ((32 - ((kib + 4) & 31)) & 31) >> 2;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        s16x4 hi, lo;
        split4<Bf16x3>(vals[t], hi, lo);
        const unsigned so = (unsigned)(8 * t * p.ldo) * 2u;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), Or, off, so, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), Or, off, so + 64u, 0);
      }
      if (tpad > 0) {
        const u32x2 z2 = {0u, 0u};
#pragma unroll
        for (int t = 0; t < 4; ++t)
          for (int g = 1; g <= tpad; ++g) {
            const unsigned so = (unsigned)(8 * t * p.ldo) * 2u;
            __builtin_amdgcn_raw_buffer_store_b64(z2, Or, off + 8u * (unsigned)g, so, 0);
            __builtin_amdgcn_raw_buffer_store_b64(z2, Or, off + 8u * (unsigned)g, so + 64u, 0);
          }
      }
    }
  };

  // chunk i of this stream is q(i) = stream + n_streams * i.  Iteration c: request chunk c + 3 (register set (c + 1) & 1),
  // contract chunk c (stage c % 3), store it, commit chunk c + 2 (requested in iteration c - 1) to stage (c + 2) % 3 — last
  // read in iteration c - 1, which every wave has left through the barrier — and meet at the barrier.
  f32x4 R0[4], R1[4];
  auto qi = [&](int i) { return stream + n_streams * i; };
  issue(qi(0), R0);
  issue(qi(1), R1);
  commit(qi(0), 0, R0);
  commit(qi(1), 1, R1);
  issue(qi(2), R0);
  __syncthreads();
  int st = 0;
  for (int c = 0; qi(c) < n_chunks; c += 2) {
    {
      issue(qi(c + 3), R1);
      const f32x16 acc = compute(st);
      epilogue(qi(c), acc);
      const int s2 = st == 0 ? 2 : st - 1;
      commit(qi(c + 2), s2, R0);
      st = st == 2 ? 0 : st + 1;
      __syncthreads();
    }
    if (qi(c + 1) >= n_chunks) break;
    {
      issue(qi(c + 4), R0);
      const f32x16 acc = compute(st);
      epilogue(qi(c + 1), acc);
      const int s2 = st == 0 ? 2 : st - 1;
      commit(qi(c + 3), s2, R1);
      st = st == 2 ? 0 : st + 1;
      __syncthreads();
    }
  }
}

int wx_grid() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8)
      n = 256;
    cus = n & ~7;
  }
  return cus;
}

}  // namespace

int csn_dev_wx = 1;      // development switch (csn_dev_set): 0 = these products on the tiled kernels of gemm_bf16x3.hip

bool csn_wx_takes(int rows, int k) { return csn_dev_wx != 0 && k == WX_K && rows > 0 && rows % 256 == 0 && rows / 256 <= 32; }


int csn_launch_wx(const CsnWxArgs& a, int out_mode, hipStream_t st) {
  if (a.n_items <= 0 || a.n_points <= 0) return 0;
  if ((a.ldx & 3) || (a.ldo & 3) || (a.n_points & 3)) return -2;
  if (out_mode == 2 && (a.tb <= 0 || (a.tb & 3))) return -2;
  if ((a.div_rows & 31) || (long long)a.n_items * ((a.n_points + WX_CH - 1) / WX_CH) + 4ll * wx_grid() >= (1ll << 31)) return -1;
  const int grid = wx_grid();
  if ((grid >> 3) < a.n_sets) return -1;
  if (out_mode == 0) hipLaunchKernelGGL((csn_wx_kernel<0>), dim3(grid), dim3(512), 0, st, a);
  else if (out_mode == 2) hipLaunchKernelGGL((csn_wx_kernel<2>), dim3(grid), dim3(512), 0, st, a);
  else return -1;
  return (int)hipGetLastError();
}
