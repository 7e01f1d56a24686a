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
#include "wx_common.h"
#include <type_traits>

namespace {


// OUT 0: fp32 map [item][rows][ldo];  OUT 2: bf16 tile planes (attn_bf16x3.hip): per row and block of tb points 16 tiles of
// [hi 32 | lo 32], block pitch 1024, row pitch ldo 16-bit elements, the padding keys of a block's last tile written as zeros
// OUT 3: xhat = LayerNorm over the 256 rows of (drop(W X) + residual), no affine (csa_models.py:115-118; the affine is applied
// where xhat is used), rstd per point, and per-stream sums of xhat over points (the pooled descriptor, csa_models.py:211-212).
// Streams take CONTIGUOUS runs of chunks here, so that the sums of an item stay in registers until the run leaves the item.
template <int OUT, bool DROP = false>
__global__ __launch_bounds__(512, 2) void csn_wx_kernel(CsnWxArgs p) {
  constexpr bool LN = OUT == 3;
  constexpr int WX_RES = LN ? 8 * WX_EB * 2 : 0;            // LN: the waves' residual blocks, 32 rows x 32 points fp32 each (32 KB)
  constexpr int WX_EBW = LN ? WX_EB / 2 : WX_EB;            // floats of a wave's epilogue block (LN: two passes of 16 rows)
  // 96 KB of stages + 32 KB of epilogue blocks; LN: 32 KB residual + 96 KB + 16 KB + 2 KB of per-wave point statistics
  // and 8 KB of running sums (one cell per lane and row group: the sums cost 6 registers at the kernel's tightest point)
  __shared__ __attribute__((aligned(16))) short smem[WX_RES + WX_NS * WX_STAGE + 8 * WX_EBW * 2 + (LN ? 1024 + 4096 : 0)];
  short* xs = smem + WX_RES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  float* eb = reinterpret_cast<float*>(smem + WX_RES + WX_NS * WX_STAGE) + wave * WX_EBW;
  float* red = reinterpret_cast<float*>(smem + WX_RES + WX_NS * WX_STAGE + 8 * WX_EBW * 2);    // LN: [wave][point][sum, m2]
  const float* resb = reinterpret_cast<const float*>(smem) + wave * WX_EB;                       // LN: the wave's residual block
  float* ps = red + 512 + wave * 256 + lane;                                                     // LN: this lane's cells ps[64 i], i < 4

  // streams: work-group b sits on XCD label b % 8; its place j = b / 8 there is (stream lane, row set)
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int spx = (int)(gridDim.x >> 3) / p.n_sets;
  if (j >= spx * p.n_sets) return;
  const int set = j % p.n_sets, n_streams = spx * 8;
  const int stream = (j / p.n_sets) * 8 + xcd;
  // chunk map.  fp32 output: chunk = 32 consecutive points of an item.  Tile planes: chunk = one 32-key tile of an attention
  // block (blocks of tb points start anywhere mod 32): a chunk then writes whole [hi 32 | lo 32] lines — chunks that straddle
  // two tiles left every 128-byte line to two work-groups, each with a partial write (K/V projection 1.98 ms against 1.63 tiled)
  constexpr bool PLANES = OUT == 2 || OUT == 4;                                  // chunk = one 32-key tile of an attention block
  const unsigned blen = PLANES ? (unsigned)p.tb : (unsigned)p.n_points;            // points per chunked block
  const unsigned cpb = (blen + WX_CH - 1) / WX_CH;                                 // chunks per block
  const unsigned cpi = cpb * ((unsigned)(p.n_points + (int)blen - 1) / blen);        // chunks per item
  const int n_chunks = p.n_items * (int)cpi;                  // (the launcher keeps it below 2^31)
  // chunks of this stream: q0, q0 + q_step, ... < q_end
  const int run = LN ? (n_chunks + n_streams - 1) / n_streams : 0;
  const int q_step = LN ? 1 : n_streams;
  const int q0 = LN ? stream * run : stream;
  const int q_end = LN ? min(q0 + run, n_chunks) : n_chunks;
  if (q0 >= q_end) return;
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
  // chunk q -> item, block, tile; first point, valid points (<= 0: a tile beyond a short last block).  A cursor walks the
  // stream's chunks with adds and compares (two 32-bit divisions per chunk and use site were a sixth of the loop's skeleton)
  struct Cursor { int q; unsigned item, blk, tile; };
  const unsigned bpi = cpi / cpb;                                                // blocks per item
  auto cursor_at = [&](int q) {
    Cursor c;
    c.q = q;
    c.item = (unsigned)q / cpi;
    const unsigned r = (unsigned)q - c.item * cpi;
    c.blk = r / cpb;
    c.tile = r - c.blk * cpb;
    return c;
  };
  const unsigned st_item = (unsigned)q_step / cpi, st_r = (unsigned)q_step - st_item * cpi;          // one step = q_step chunks
  const unsigned st_blk = st_r / cpb, st_tile = st_r - st_blk * cpb;
  auto advance = [&](Cursor& c) {
    c.q += q_step;
    c.tile += st_tile;
    if (c.tile >= cpb) { c.tile -= cpb; ++c.blk; }
    c.blk += st_blk;
    if (c.blk >= bpi) { c.blk -= bpi; ++c.item; }
    c.item += st_item;
  };
  auto locate = [&](const Cursor& c, int& col0, int& valid) {
    col0 = (int)(c.blk * blen + c.tile * WX_CH);
    valid = min(min(WX_CH, (int)blen - (int)(c.tile * WX_CH)), p.n_points - col0);
  };
  // (request and commit are unconditional — a chunk beyond the stream's last one is requested with every lane switched off and
  //  committed as zeros: with a branch around either, the compiler can no longer pair a request with its commit and drains
  //  the whole memory queue, this chunk's stores and the next request included, before it touches a staging register:
  //  s_waitcnt vmcnt(0) at the top of every iteration, 0.64 ms for the Q projection instead of 0.4x)
  auto issue = [&](const Cursor& cu, f32x4* R) {
    const bool exists = cu.q < q_end;
    int col0, valid;
    locate(cu, col0, valid);
    if (valid < 0 || !exists || (p.ablate & 4)) valid = 0;
    const unsigned item = exists ? cu.item : 0u;
    if (!exists) col0 = 0;
    const u32x4 Xr = wx_rsrc(p.x + (long long)item * p.x_item_stride + col0, ((long long)(WX_K - 1) * p.ldx + valid) * 4);
    const unsigned off = 4 * c4 < valid ? x_voff : CSN_OOB;
#pragma unroll
    for (int i = 0; i < 4; ++i) wx_request(R[i], Xr, off, (unsigned)(64 * i * p.ldx) * 4u);
  };
  auto commit = [&](int stage, const f32x4* R) {
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
    if (p.ablate & 1) {
      acc[0] = __builtin_bit_cast(float, (int)xh[0]);
      return acc;
    }
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
  auto epilogue = [&](const Cursor& cu, f32x16 acc) {
    const unsigned item = cu.item, blk = cu.blk, tile = cu.tile;
    int col0, valid;
    locate(cu, col0, valid);
    const int n = col0 + 4 * c8;
    if (p.ablate & 2) valid = 0;
    const bool n_ok = 4 * c8 < valid;                                  // (a tile beyond a short last block: no lane stores)
    if (true_div) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = acc[r] / p.div_val;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) eb[csn_acc_row(r, h) * 32 + l31] = acc[r] * dscale;
    // OUT 4: the first n_f32 row sets leave as fp32 maps (out_f32), the others as tile planes (out), from the same chunks of x
    const bool f32_set = OUT == 0 || (OUT == 4 && set < p.n_f32);
    const int pset = OUT == 4 ? set - p.n_f32 : set;
    if (f32_set) {
      f32x4 vals[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) vals[t] = *reinterpret_cast<const f32x4*>(&eb[(erow + 8 * t) * 32 + 4 * c8]);
      float* const ob = OUT == 4 ? p.out_f32 : reinterpret_cast<float*>(p.out);
      const long long ois = OUT == 4 ? p.out_f32_item_stride : p.out_item_stride;
      const int ol = OUT == 4 ? p.ldo_f32 : p.ldo;
      const csn_rsrc_t Or = csn_make_rsrc(ob + (long long)item * ois + (long long)(256 * set) * ol, (long long)256 * ol * 4);
      const unsigned off = n_ok ? (unsigned)((32 * wave + erow) * ol + n) * 4u : CSN_OOB;
#pragma unroll
      for (int t = 0; t < 4; ++t) csn_bstore4(vals[t], Or, off, (unsigned)(8 * t * ol) * 4u);
    } else {
      // the tile's 32 keys leave whole, 8 keys = 16 bytes per lane and plane (lane -> row lane / 4 + 16 t, keys 8 (lane % 4) ..):
      // points beyond the block's (or the row's) end were never loaded, their sums are zeros — exactly the padding the
      // attention kernels expect behind a block's last key
// (-DCSN_WX_PLANE_LINES=1: every store instruction writes 8 whole 128-byte lines instead of 16 half lines, at twice the
//  conversion work — measured 1.10 ms against 1.06 for the K / V projection, same bits; off)
#ifndef CSN_WX_PLANE_LINES
#define CSN_WX_PLANE_LINES 0
#endif
      const csn_rsrc_t Or = csn_make_rsrc(reinterpret_cast<short*>(p.out) + (long long)item * p.out_item_stride + (long long)(256 * pset) * p.ldo,
                                          (long long)256 * p.ldo * 2);
      if constexpr (CSN_WX_PLANE_LINES != 0) {
        // whole 128-byte lines per store instruction: lane -> row lane / 8 + 8 t, 16-byte unit lane % 8 of the row's [hi 32 | lo 32]
        // (units 0..3: keys 8 u .. + 7 of the hi plane, 4..7: of the lo plane).  A lane splits its 8 keys and keeps one plane —
        // twice the conversion work of the form below, where a store instruction wrote 16 half lines (rows 40 KB apart)
        const int prow = lane >> 3, u = lane & 7, ku = u & 3;
        const unsigned tcol = blk * 1024u + tile * 64u + 8u * (unsigned)u;
        const unsigned off = valid > 0 ? ((unsigned)((32 * wave + prow) * p.ldo) + tcol) * 2u : CSN_OOB;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(&eb[(prow + 8 * t) * 32 + 8 * ku]);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(&eb[(prow + 8 * t) * 32 + 8 * ku + 4]);
          s16x4 h0, l0, h1, l1;
          split4<Bf16x3>(v0, h0, l0);
          split4<Bf16x3>(v1, h1, l1);
          const s16x8 keep = u < 4 ? join8(h0, h1) : join8(l0, l1);
          csn_bstore4(__builtin_bit_cast(f32x4, keep), Or, off, (unsigned)(8 * t * p.ldo) * 2u);
        }
      } else {
        const int prow = lane >> 2, c4k = lane & 3;
        const unsigned tcol = blk * 1024u + tile * 64u + 8u * (unsigned)c4k;
        const unsigned off = valid > 0 ? ((unsigned)((32 * wave + prow) * p.ldo) + tcol) * 2u : CSN_OOB;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(&eb[(prow + 16 * t) * 32 + 8 * c4k]);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(&eb[(prow + 16 * t) * 32 + 8 * c4k + 4]);
          s16x4 h0, l0, h1, l1;
          split4<Bf16x3>(v0, h0, l0);
          split4<Bf16x3>(v1, h1, l1);
          const unsigned so = (unsigned)(16 * t * p.ldo) * 2u;
          csn_bstore4(__builtin_bit_cast(f32x4, join8(h0, h1)), Or, off, so);
          csn_bstore4(__builtin_bit_cast(f32x4, join8(l0, l1)), Or, off, so + 64u);
        }
      }
    }
  };

  // ---- LayerNorm epilogue (OUT 3) --------------------------------------------------------------------------------------
  // residual of a chunk: the wave's 32 rows x 32 points, 4 requests of 8 rows (lane -> row lane / 8, piece lane % 8) that land
  // row-major in the wave's residual block one iteration after they were asked for
  const unsigned res_lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((short __attribute__((address_space(3)))*)smem))
                           + (unsigned)wave * (WX_EB * 4u);
  auto issue_res = [&](const Cursor& cu) __attribute__((always_inline)) {
    const bool exists = cu.q < q_end;
    int col0, valid;
    locate(cu, col0, valid);
    if (valid < 0 || !exists || (p.ablate & 4)) valid = 0;
    const unsigned item = __builtin_amdgcn_readfirstlane(exists ? cu.item : 0u);
    if (!exists) col0 = 0;
    const long long rs = p.res_index ? (long long)wx_sload(p.res_index + item) : (long long)item;
    const u32x4 Rr = wx_rsrc(p.res + rs * p.res_shape_stride + (long long)(32 * wave) * p.ldo + col0, ((long long)31 * p.ldo + valid) * 4);
    const unsigned off = 4 * c8 < valid ? (unsigned)(erow * p.ldo + 4 * c8) * 4u : CSN_OOB;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (the block's previous chunk has been read)
#pragma unroll
    for (int t = 0; t < 4; ++t) wx_dma(res_lds + 1024u * t, Rr, off, (unsigned)(8 * t * p.ldo) * 4u);
  };
  constexpr int NST = LN ? 5 : 4;                      // stores of one epilogue (always issued; switched-off lanes still count)
  const unsigned thr16 = csn_drop_threshold16(p.dropout_p);
  const float keep_scale = DROP ? 1.f / (1.f - p.dropout_p) : 1.f;
  // sums of xhat over the points of the item the run is in: lane partials (row lane / 8 + 8 t + 16 half, points 4 (lane % 8)..)
  if constexpr (LN) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ps[64 * i] = 0.f;
  }
  unsigned sum_item = 0;
  auto flush_sums = [&]() __attribute__((always_inline)) {
    if (!p.sum_ws) return;
    const unsigned slot = (unsigned)stream - (sum_item * cpi) / (unsigned)run;
    float* dst = p.sum_ws + ((long long)sum_item * p.sum_slots + slot) * 256 + 32 * wave + erow;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = ps[64 * i];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      if (c8 == 0) dst[16 * (i >> 1) + 8 * (i & 1)] = v;
      ps[64 * i] = 0.f;
    }
  };
  auto epilogue_ln = [&](const Cursor& cu, f32x16 acc) __attribute__((always_inline)) {
    int col0, valid;
    locate(cu, col0, valid);
    if (p.ablate & 2) valid = 0;
    const bool pt_ok = l31 < valid;
    if (cu.item != sum_item) {                          // (work-group uniform)
      flush_sums();
      sum_item = cu.item;
    }
    // fc dropout in the accumulator layout: rows r, r + 1 (r even) are one channel pair of point col0 + l31 — one hash
    if constexpr (DROP) {
      const unsigned salt = csn_block_salt((unsigned long long)cu.item, p.seed);
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const unsigned hp = csn_fc_pair(32 * wave + csn_acc_row(r, h), (unsigned)p.ldo, (unsigned)(col0 + l31), salt);
        acc[r] = csn_keep16(hp, 0, thr16) ? acc[r] * keep_scale : 0.f;
        acc[r + 1] = csn_keep16(hp, 1, thr16) ? acc[r + 1] * keep_scale : 0.f;
        if (r & 2) __builtin_amdgcn_sched_barrier(0);   // (two hashes at a time: eight at once cost nine more registers)
      }
    }
    // + residual: requested one iteration ago; behind that request only the stores of that iteration and this one's chunk request
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NST + 4) : "memory");
#pragma unroll
    for (int g = 0; g < 4; ++g) {                       // (in fours: sixteen reads in flight at once cost sixteen more registers)
#pragma unroll
      for (int r = 4 * g; r < 4 * g + 4; ++r) acc[r] += resb[csn_acc_row(r, h) * 32 + l31];
      __builtin_amdgcn_sched_barrier(0);
    }
    // statistics of the point over the wave's 32 rows (sum, squares about the wave's mean), combined over the 8 waves below
    float s1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s1 += acc[r];
    s1 += csn_xhalf(s1);
    const float mw = s1 * (1.f / 32.f);
    float m2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) m2 += (acc[r] - mw) * (acc[r] - mw);
    m2 += csn_xhalf(m2);
    if (h == 0) *reinterpret_cast<float2*>(&red[(wave * 32 + l31) * 2]) = make_float2(s1, m2);
    {
      Cursor cn = cu;
      advance(cn);
      issue_res(cn);                                    // the next chunk's residual (the block's reads are done: values in use)
    }
    __syncthreads();
    // (Chan's combination of the waves' (sum, squares about own mean); two sweeps over the 8 entries keep 16 registers free)
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) tot += red[(w * 32 + l31) * 2];
    const float mean = tot * (1.f / 256.f);
    float var = 0.f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const float2 sw = *reinterpret_cast<const float2*>(&red[(w * 32 + l31) * 2]);
      const float dm = sw.x * (1.f / 32.f) - mean;
      var += sw.y + 32.f * dm * dm;
    }
    const float rstd = 1.f / sqrtf(var * (1.f / 256.f) + p.eps);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = pt_ok ? (acc[r] - mean) * rstd : 0.f;
    {
      const csn_rsrc_t Sr = csn_make_rsrc(p.rstd + (long long)cu.item * p.n_points, (long long)p.n_points * 4);
      csn_bstore(rstd, Sr, (wave == 0 && h == 0 && pt_ok) ? (unsigned)(col0 + l31) * 4u : CSN_OOB);
    }
    // out through the wave's block in two passes of 16 rows, 16-byte row stores
    const int n = col0 + 4 * c8;
    const bool n_ok = 4 * c8 < valid;
    const u32x4 Or = wx_rsrc(reinterpret_cast<float*>(p.out) + (long long)cu.item * p.out_item_stride, (long long)256 * p.ldo * 4);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) eb[csn_acc_row(rr, h) * 32 + l31] = acc[8 * half + rr];
      const unsigned off = n_ok ? (unsigned)((32 * wave + 16 * half + erow) * p.ldo + n) * 4u : CSN_OOB;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&eb[(erow + 8 * t) * 32 + 4 * c8]);
        wx_store4(v, Or, off, (unsigned)(8 * t * p.ldo) * 4u);
        ps[64 * (2 * half + t)] += (v[0] + v[1]) + (v[2] + v[3]);
      }
    }
  };

  // Chunk i of this stream is stream + n_streams * i.  Iteration c: request chunk c + 3 (register set (c + 1) & 1),
  // contract chunk c (stage c % 3) | store it, commit chunk c + 2 (requested in iteration c - 1) to stage (c + 2) % 3.
  // Staggered halves (p.stagger; measured: no gain — the CU's memory pipe, not the overlap of the parts, sets the pace — and
  // off by default): a barrier between the two parts, and waves 4..7 — the SIMD partners of waves 0..3 — run one
  // part behind (one extra barrier in front of their loop, one behind the loop of the others): a SIMD then always has one wave
  // in its matrix part beside one in its memory part, instead of eight waves in lock step leaving each pipe idle in turn.
  // Hazards (barrier intervals; waves 0..3 contract chunk c in 2c and commit in 2c + 1, waves 4..7 in 2c + 1 and 2c + 2):
  // chunk c lands in stage c % 3 in intervals 2c - 3 / 2c - 2 and is first read in 2c; the stage's previous chunk c - 3 was
  // last read in 2c - 5.  Without the stagger: one barrier per iteration, written in c - 2, read in c, previous read in c - 3.
  // (p.ablate, development timing only: 1 no matrix instructions, 2 every store lane off, 4 every request lane off)
  // THREE chunks in flight: chunk k travels in register set k % 3 and lands in LDS stage k % 3.  Iteration c requests chunk
  // c + 4, contracts and stores chunk c, then commits chunk c + 2 — requested at the top of iteration c - 2, so a request has
  // two whole iterations (and a store three) to complete: 96 KB of loads and 96 KB of stores per CU in flight.  (With two
  // sets — 64 KB — loads-only and stores-only timings added up to the kernel's time: each was bound by its own queue depth
  // times the memory latency, ~3 us under load.)
  f32x4 R0[4], R1[4], R2[4];
  const bool stag = p.stagger != 0, late = stag && wave >= 4;
  // issued behind a request when its commit waits for it: three epilogues' stores, two requests (LN: three residual requests)
  constexpr int BEHIND = 3 * NST + 8 + (LN ? 12 : 0);
  Cursor ci = cursor_at(q0), ce = ci;              // request cursor (runs four chunks ahead), contraction / store cursor
  // the loop waits for a request by counting what was issued behind it.  In front of the first iterations there are no
  // epilogues yet: NST stores through an empty window (dropped by the range check, counted like any other) stand in for each,
  // so that ONE count holds for every iteration (two wait statements on two branches made the compiler copy the in-flight
  // registers between them)
  auto standin = [&](auto count) {
    const u32x4 none = wx_rsrc(nullptr, 0);
    const unsigned oob = CSN_OOB;
#pragma unroll
    for (int i = 0; i < decltype(count)::value; ++i)         // (asm: identical stores through a builtin are merged into one)
      asm volatile("buffer_store_dword %0, %0, %1, 0 offen" :: "v"(oob), "s"(none) : "memory");
  };
  issue(ci, R0); advance(ci);
  issue(ci, R1); advance(ci);
  wx_arrived<0>(R0);
  wx_arrived<0>(R1);
  commit(0, R0);
  commit(1, R1);
  issue(ci, R2); advance(ci);                          // chunk 2
  standin(std::integral_constant<int, NST + (LN ? 4 : 0)>{});          // (LN: a residual request sits in front of each epilogue's stores)
  issue(ci, R0); advance(ci);                          // chunk 3
  if constexpr (LN) {
    sum_item = ce.item;
    issue_res(ce);                                     // residual of chunk 0
  }
  standin(std::integral_constant<int, NST>{});
  __syncthreads();
  if (late) __syncthreads();
  // one iteration: ST = c % 3 (compile time: the loop is unrolled by three), RQ the set chunk c + 4 goes to, RC the set of chunk c + 2
  auto iteration = [&](auto st_c, f32x4* RQ, f32x4* RC) {
    constexpr int ST = decltype(st_c)::value;
    issue(ci, RQ); advance(ci);
    const f32x16 acc = compute(ST);
    if (stag) __syncthreads();
    if constexpr (LN) epilogue_ln(ce, acc);
    else epilogue(ce, acc);
    advance(ce);
    wx_arrived<BEHIND>(RC);
    commit((ST + 2) % 3, RC);
    __syncthreads();
  };
  while (true) {
    if (ce.q >= q_end) break;
    iteration(std::integral_constant<int, 0>{}, R1, R2);
    if (ce.q >= q_end) break;
    iteration(std::integral_constant<int, 1>{}, R2, R0);
    if (ce.q >= q_end) break;
    iteration(std::integral_constant<int, 2>{}, R0, R1);
  }
  if constexpr (LN) flush_sums();
  if (stag && !late) __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (requests beyond the last chunk: every lane off, nothing fetched)
}


// out[item][c] = sum of the item's slots of sum_ws[item][slots][256] (fp64 accumulation; the slots the launch geometry wrote)
__global__ void csn_wx_ln_sums_kernel(const float* __restrict__ ws, float* __restrict__ out, int cpi, int run, int slots) {
  const long long e = blockIdx.x;
  const int first = (int)((e * cpi) / run), last = (int)(((e + 1) * cpi - 1) / run);
  const float* src = ws + e * slots * 256 + threadIdx.x;
  double s = 0.0;
  for (int k = 0; k <= last - first; ++k) s += (double)src[k * 256];
  out[e * 256 + threadIdx.x] = (float)s;
}

void wx_ln_geometry(int n_items, int n_points, int& cpi, int& run, int& slots) {
  cpi = (n_points + WX_CH - 1) / WX_CH;
  const long long n_chunks = (long long)n_items * cpi;
  run = (int)((n_chunks + wx_grid() - 1) / wx_grid());
  if (run < 1) run = 1;
  slots = (cpi + run - 1) / run + 1;
}

}  // namespace

// LayerNorm backward + dCtx alternate over groups of this many evaluations (0: one launch each).  Measured over the config-3
// step, 256 evaluations, with the two separate launches (profiles/r4aa_ln_backward_groups.txt): groups of 8 +0.68 ms, 16 +0.25,
// 32 -0.08, 64 -0.10..-0.18, 128 -0.14 — not cache residency (a group of 128 is 1.3 GB of dz); with the fused kernel
// (wx_lnb.hip) one launch is best (25.96 against 26.04 in groups of 128).  The same bits either way.
int csn_dev_lnb_group = 0;
// development switch (csn_dev_set): bit 0 these products on the streaming kernel (0: on the tiled kernels of gemm_bf16x3.hip), bit 1
// staggered wave halves, bit 2 out-projection + LayerNorm back on the tiled kernel, bit 3 LayerNorm backward fused into the dCtx
// stream (wx_lnb.hip), bits 4..7 timing-only ablations
int csn_dev_wx = 9;

bool csn_wx_takes(int rows, int k) { return (csn_dev_wx & 1) != 0 && k == WX_K && rows > 0 && rows % 256 == 0 && rows / 256 <= 32; }


// the launch geometry the persistent grid and its 32-bit chunk cursors cover (otherwise: CSN_NOT_TAKEN, the tiled kernels)
bool csn_wx_geometry_takes(int n_items, int n_points, int n_sets) {
  return ((long long)n_items * ((n_points + WX_CH - 1) / WX_CH + 16) + 4ll * wx_grid()) * 2 < (1ll << 31) && (wx_grid() >> 3) >= n_sets;
}

int csn_wx_ln_sum_slots(int n_items, int n_points) {
  int cpi, run, slots;
  wx_ln_geometry(n_items, n_points, cpi, run, slots);
  return slots;
}

int csn_launch_wx_ln_sums(const float* ws, float* out, int n_items, int n_points, hipStream_t st) {
  if (n_items <= 0) return 0;
  int cpi, run, slots;
  wx_ln_geometry(n_items, n_points, cpi, run, slots);
  hipLaunchKernelGGL(csn_wx_ln_sums_kernel, dim3(n_items), dim3(256), 0, st, ws, out, cpi, run, slots);
  return (int)hipGetLastError();
}

int csn_launch_wx(const CsnWxArgs& a, int out_mode, hipStream_t st) {
  if (a.n_items <= 0 || a.n_points <= 0) return 0;
  if ((a.ldx & 3) || (a.ldo & 3) || (a.n_points & 3)) return -2;
  if ((out_mode == 2 || out_mode == 4) && (a.tb <= 0 || (a.tb & 3))) return -2;
  if (out_mode == 4 && (a.n_f32 <= 0 || a.n_f32 >= a.n_sets || !a.out_f32 || (a.ldo_f32 & 3))) return -2;
  if ((a.div_rows & 31) || !csn_wx_geometry_takes(a.n_items, a.n_points, a.n_sets)) return CSN_NOT_TAKEN;
  const int grid = wx_grid();
  CsnWxArgs b = a;
  b.stagger = (csn_dev_wx & 2) ? 1 : 0;
  b.ablate = (csn_dev_wx >> 4) & 15;
  if (out_mode == 3) {
    if (a.n_sets != 1 || !a.res || !a.rstd || a.div_rows) return -2;
    b.stagger = 0;
    if (a.sum_ws && a.sum_slots != csn_wx_ln_sum_slots(a.n_items, a.n_points)) return -2;
    if (a.dropout_p > 0.f) hipLaunchKernelGGL((csn_wx_kernel<3, true>), dim3(grid), dim3(512), 0, st, b);
    else hipLaunchKernelGGL((csn_wx_kernel<3, false>), dim3(grid), dim3(512), 0, st, b);
  } else if (out_mode == 0) hipLaunchKernelGGL((csn_wx_kernel<0>), dim3(grid), dim3(512), 0, st, b);
  else if (out_mode == 2) hipLaunchKernelGGL((csn_wx_kernel<2>), dim3(grid), dim3(512), 0, st, b);
  else if (out_mode == 4) hipLaunchKernelGGL((csn_wx_kernel<4>), dim3(grid), dim3(512), 0, st, b);
  else return CSN_NOT_TAKEN;
  return (int)hipGetLastError();
}
