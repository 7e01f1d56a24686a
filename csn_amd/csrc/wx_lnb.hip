// LayerNorm backward fused into the dCtx stream (backward of MID-FC/csa_models.py:115-118, bf16x3 mode, d_model = d_inner = 256):
//   dx   = scale[e][c] dfeats[e / group][c][n] (dense evaluations) + rows[e][c]                 the gradient w.r.t. xhat
//   dz   = rstd (dx - mean_c dx - xhat mean_c(dx xhat))                                          -> dz_res (the residual branch)
//   dz  *= fc dropout mask / keep                                                                -> dz (read again by the W_fc gradient)
//   dCtx = W_fc^T dz                                                                             -> dctx
// One pass: xhat is read once, dz never comes back from memory for the product (csn_ln_bwd_kernel + csn_wx_kernel<0> read
// xhat, write dz, read dz, write dCtx).  The skeleton is wx_stream.hip's: persistent work-groups, W_fc^T stationary in registers
// as A fragments, 32-point chunks, three LDS stages of hi / lo planes, three chunks of xhat in flight in registers — but the
// chunk that is committed to LDS is COMPUTED on its way in:
//   * staging rows of a thread are channel PAIRS (2 r, 2 r + 1, 128 + 2 r, 129 + 2 r; r = tid / 8): one dropout hash decides a
//     pair (csn_common.h), so a thread hashes 2 x 4 points per chunk instead of 16 elements;
//   * the dfeats chunk comes by LDS-DMA one chunk ahead (every thread reads back its own 64 bytes: no barrier, its own vmcnt);
//   * the two row sums of a point over the 256 channels: 4 rows in the thread, 8 row groups of the wave by cross-lane adds, 8
//     waves through 2 KB of LDS and ONE extra barrier per chunk.
// Wait counts (hand-counted, see wx_stream.hip): the commit of chunk c + 2 waits for the dfeats request issued inside the commit
// of the previous iteration; behind it: that commit's 4 (8 with dz_res) stores, this iteration's chunk request (4 + 1 loads)
// and the 4 stores of its epilogue.
#include "csn_common.h"
#include "csn_kernels.h"
#include "wx_common.h"
#include <type_traits>

namespace {

constexpr int LNB_G = 8 * WX_EB * 2;            // 16-bit elements of the dfeats chunk buffer (32 KB)
constexpr int LNB_EBW = WX_EB / 2;              // floats of a wave's epilogue block (two passes of 16 rows)

template <int N>
CSN_DEVINL void lnb_arrived(f32x4* R) {
  asm volatile("s_waitcnt vmcnt(%5)" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]) : "n"(N) : "memory");
}

// v summed over the 8 lanes l ^ {8, 16, 32} (the wave's 8 row groups of one point column), in every one of them — on the vector
// pipe: a rotation by 8 inside the 16-lane rows, then the row-pair and half-wave swaps of gfx950 (three ds_bpermute per value
// — 24 per chunk and thread — were LDS traffic and 3 x ~100 cycles of latency in a row)
CSN_DEVINL float lnb_sum_row_groups(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
  // (asm, not __builtin_amdgcn_permlane16_swap / 32_swap: hipcc 7.2 maps BOTH elements of the builtin's result to the first
  //  register — `v_permlane16_swap v5, v4 ; v_add_f32 v2, v5, v5` — and the sum comes out as twice one half)
  float c = v;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v), "+v"(c));
  v += c;
  c = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v), "+v"(c));
  return v + c;
}

template <bool DROP, bool RES>
__global__ __launch_bounds__(512, 2) void csn_wx_lnb_kernel(CsnWxLnbArgs p) {
  // [dfeats chunk 32 KB | 3 stages 96 KB | 8 epilogue blocks 16 KB | row sums 2 KB | (scale, row constant) of the item's channels 2 KB]
  __shared__ __attribute__((aligned(16))) short smem[LNB_G + WX_NS * WX_STAGE + 8 * LNB_EBW * 2 + 1024 + 1024];
  short* xs = smem + LNB_G;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  float* eb = reinterpret_cast<float*>(smem + LNB_G + WX_NS * WX_STAGE) + wave * LNB_EBW;
  float* red = reinterpret_cast<float*>(smem + LNB_G + WX_NS * WX_STAGE + 8 * LNB_EBW * 2);      // [wave][sum | sum x][32 points]
  const float* gbuf = reinterpret_cast<const float*>(smem) + 4 * tid;                              // this thread's piece i at + 2048 i
  float* ctab = red + 512;                                                                         // [channel][scale, row constant]

  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int n_streams = (int)gridDim.x;
  const int stream = j * 8 + xcd;
  const unsigned cpi = (unsigned)(p.n_points + WX_CH - 1) / WX_CH;                 // chunks per item
  const int n_chunks = p.n_items * (int)cpi;
  const int run = (n_chunks + n_streams - 1) / n_streams;                          // contiguous runs: an item's constants change rarely
  const int q0 = stream * run, q_end = min(q0 + run, n_chunks);
  if (q0 >= q_end) return;

  // W_fc^T: the wave's 32 rows (of d_inner) as A fragments
  s16x8 Wh[WX_K / 16], Wl[WX_K / 16];
  {
    const float* wrow = p.w + ((long long)(32 * wave + l31) * WX_K + 8 * h);
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

  // staging: thread -> channel rows 2 r + (i & 1) + 128 (i >> 1), r = tid / 8; points 4 (tid % 8) .. + 3 of the chunk
  const int krow = tid >> 3, c4 = tid & 7;
  const int ld = p.ld;
  const unsigned x_voff = (unsigned)(2 * krow * ld + 4 * c4) * 4u;
  auto row_soff = [&](int i) { return (unsigned)(((i & 1) + 128 * (i >> 1)) * ld) * 4u; };
  struct Cursor { int q; unsigned item, tile; };
  auto cursor_at = [&](int q) { Cursor c; c.q = q; c.item = (unsigned)q / cpi; c.tile = (unsigned)q - c.item * cpi; return c; };
  auto advance = [&](Cursor& c) { ++c.q; if (++c.tile >= cpi) { c.tile = 0; ++c.item; } };
  auto locate = [&](const Cursor& c, int& col0, int& valid, unsigned& item) {
    const bool exists = c.q < q_end;
    col0 = exists ? (int)(c.tile * WX_CH) : 0;
    valid = exists && !(p.ablate & 4) ? min(WX_CH, p.n_points - col0) : 0;
    item = __builtin_amdgcn_readfirstlane(exists ? c.item : 0u);
  };
  // a chunk of xhat (4 pieces) and its 4 rstd values into a register set of 5
  auto issue_x = [&](const Cursor& cu, f32x4* R) {
    int col0, valid; unsigned item;
    locate(cu, col0, valid, item);
    const u32x4 Xr = wx_rsrc(p.xhat + (long long)(p.e_base + item) * p.eval_stride + col0, ((long long)(WX_K - 1) * ld + valid) * 4);
    const unsigned off = 4 * c4 < valid ? x_voff : CSN_OOB;
#pragma unroll
    for (int i = 0; i < 4; ++i) wx_request(R[i], Xr, off, row_soff(i));
    const u32x4 Sr = wx_rsrc(p.rstd + (long long)(p.e_base + item) * p.n_points + col0, (long long)valid * 4);
    wx_request(R[4], Sr, 4 * c4 < valid ? (unsigned)(16 * c4) : CSN_OOB, 0u);
  };
  // the dfeats chunk of the same rows and points, straight into LDS: thread t's piece i lands at 8192 i + 16 t
  const unsigned g_lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((short __attribute__((address_space(3)))*)smem)) + (unsigned)wave * 1024u;
  auto issue_g = [&](const Cursor& cu) {
    int col0, valid; unsigned item;
    locate(cu, col0, valid, item);
    const unsigned e = (unsigned)p.e_base + item;
    const bool dense = (int)e < p.n_dense;
    const unsigned src = p.dxhat_group > 1 ? e / (unsigned)p.dxhat_group : e;
    const u32x4 Gr = wx_rsrc(dense ? p.dxhat + (long long)src * p.eval_stride + col0 : nullptr, dense ? ((long long)(WX_K - 1) * ld + valid) * 4 : 0);
    const unsigned off = 4 * c4 < valid ? x_voff : CSN_OOB;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (the buffer's previous chunk has been read)
#pragma unroll
    for (int i = 0; i < 4; ++i) wx_dma(g_lds + 8192u * i, Gr, off, row_soff(i));
  };

  // B fragments of the contraction: as in wx_stream.hip (k rows are channels)
  const int grp = lane >> 4, gq = (lane >> 2) & 3, gp = lane & 3;
  const int tr_base = (8 * (grp >> 1) + gq) * WX_CH + 16 * (grp & 1) + 4 * gp;
  auto compute = [&](int stage) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const short* xh = xs + stage * WX_STAGE + tr_base;
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
  // dCtx chunk out: the wave's 32 x 32 block through its LDS block in two passes of 16 rows, 16-byte row stores (4 stores)
  const int erow = lane >> 3, c8 = lane & 7;
  auto epilogue = [&](const Cursor& cu, const f32x16& acc) {
    int col0, valid; unsigned item;
    locate(cu, col0, valid, item);
    if (p.ablate & 2) valid = 0;
    const bool n_ok = 4 * c8 < valid;
    const u32x4 Or = wx_rsrc(p.dctx + (long long)(p.e_base + item) * p.dctx_eval_stride, (long long)256 * ld * 4);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) eb[csn_acc_row(rr, h) * 32 + l31] = acc[8 * half + rr];
      const unsigned off = n_ok ? (unsigned)((32 * wave + 16 * half + erow) * ld + col0 + 4 * c8) * 4u : CSN_OOB;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&eb[(erow + 8 * t) * 32 + 4 * c8]);
        wx_store4(v, Or, off, (unsigned)(8 * t * ld) * 4u);
      }
    }
  };

  // ---- the commit: LayerNorm backward of a chunk on its way into LDS -------------------------------------------------------
  const unsigned thr16 = csn_drop_threshold16(p.dropout_p);
  const float keep_scale = DROP ? 1.f / (1.f - p.dropout_p) : 1.f;
  unsigned cst_item = 0xffffffffu;                           // the item whose channel constants sit in ctab
  auto commit = [&](auto wait_c, int stage, f32x4* R, const Cursor& cu, const Cursor& cg) __attribute__((always_inline)) {
    int col0, valid; unsigned item;
    locate(cu, col0, valid, item);
    if (p.ablate & 2) valid = 0;
    const unsigned e = (unsigned)p.e_base + item;
    const bool dense = (int)e < p.n_dense;
    if (item != cst_item) {                                  // (work-group uniform; once per item and stream: every wave is past the
      cst_item = item;                                       //  barrier that ended the last commit, nobody reads the old table)
      if (tid < WX_K) {
        const float s = (dense && p.dxhat_scale) ? p.dxhat_scale[(long long)e * WX_K + tid] : 1.f;
        const float r = p.dxhat_rows ? p.dxhat_rows[(long long)e * WX_K + tid] : 0.f;
        *reinterpret_cast<float2*>(&ctab[2 * tid]) = make_float2(s, r);
      }
      __syncthreads();
    }
    lnb_arrived<decltype(wait_c)::value>(R);
    float sc[4], rw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float2 c2 = *reinterpret_cast<const float2*>(&ctab[2 * (2 * krow + (i & 1) + 128 * (i >> 1))]);
      sc[i] = c2.x;
      rw[i] = c2.y;
    }
    f32x4 gx[4];
    f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gbuf + 2048 * i);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        gx[i][jj] = dense ? fmaf(g[jj], sc[i], rw[i]) : rw[i];
        s1[jj] += gx[i][jj];
        s2[jj] += gx[i][jj] * R[i][jj];
      }
    }
    // the wave's 8 row groups (lane bits 3..5), then the 8 waves through LDS
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      s1[jj] = lnb_sum_row_groups(s1[jj]);
      s2[jj] = lnb_sum_row_groups(s2[jj]);
    }
    if (lane < 8) {
      *reinterpret_cast<f32x4*>(&red[(wave * 2 + 0) * 32 + 4 * c4]) = s1;
      *reinterpret_cast<f32x4*>(&red[(wave * 2 + 1) * 32 + 4 * c4]) = s2;
    }
    {
      Cursor cn = cg;
      issue_g(cn);                                           // the dfeats chunk of the NEXT commit (this one's has been read)
    }
    __syncthreads();
    f32x4 m1 = f32x4{0.f, 0.f, 0.f, 0.f}, m2 = m1;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      m1 += *reinterpret_cast<const f32x4*>(&red[(w * 2 + 0) * 32 + 4 * c4]);
      m2 += *reinterpret_cast<const f32x4*>(&red[(w * 2 + 1) * 32 + 4 * c4]);
    }
    m1 *= (1.f / WX_K);
    m2 *= (1.f / WX_K);
    const f32x4 rs = R[4];
    const long long ebase = (long long)e * p.eval_stride;
    const u32x4 Zr = wx_rsrc(p.dz + ebase + col0, ((long long)(WX_K - 1) * ld + valid) * 4);
    const u32x4 Zres = wx_rsrc(RES ? p.dz_res + ebase + col0 : nullptr, RES ? ((long long)(WX_K - 1) * ld + valid) * 4 : 0);
    const unsigned off = 4 * c4 < valid ? x_voff : CSN_OOB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) gx[i][jj] = rs[jj] * (gx[i][jj] - m1[jj] - R[i][jj] * m2[jj]);
      if constexpr (RES) wx_store4(gx[i], Zres, off, row_soff(i));
    }
    if constexpr (DROP) {
      const unsigned salt = csn_block_salt((unsigned long long)e, p.seed);
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const unsigned hp = csn_fc_pair(2 * krow + 128 * pr, (unsigned)ld, (unsigned)(col0 + 4 * c4 + jj), salt);
          gx[2 * pr][jj] = csn_keep16(hp, 0, thr16) ? gx[2 * pr][jj] * keep_scale : 0.f;
          gx[2 * pr + 1][jj] = csn_keep16(hp, 1, thr16) ? gx[2 * pr + 1][jj] * keep_scale : 0.f;
        }
    }
    short* dst = xs + stage * WX_STAGE + 2 * krow * WX_CH + 4 * c4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wx_store4(gx[i], Zr, off, row_soff(i));
      s16x4 hi, lo;
      split4<Bf16x3>(gx[i], hi, lo);
      const int ro = ((i & 1) + 128 * (i >> 1)) * WX_CH;
      *reinterpret_cast<s16x4*>(dst + ro) = hi;
      *reinterpret_cast<s16x4*>(dst + ro + WX_PLANE) = lo;
    }
  };

  // ---- pipeline ---------------------------------------------------------------------------------------------------------------
  // chunk k travels in register set k % 3 and lands in stage k % 3.  Iteration c: request chunk c + 4, contract and store chunk
  // c, commit chunk c + 2 (its xhat requested in iteration c - 2, its dfeats at the end of iteration c - 1).
  f32x4 R0[5], R1[5], R2[5];
  Cursor ci = cursor_at(q0), ce = ci, cc = ci, cg = ci;       // request, contraction, commit, dfeats-request cursors
  issue_x(ci, R0); advance(ci);
  issue_x(ci, R1); advance(ci);
  issue_g(cg); advance(cg);                                   // dfeats of chunk 0
  commit(std::integral_constant<int, 0>{}, 0, R0, cc, cg); advance(cc); advance(cg);      // (requests dfeats of chunk 1)
  issue_x(ci, R2); advance(ci);                               // chunk 2
  issue_x(ci, R0); advance(ci);                               // chunk 3
  __syncthreads();                                            // (the row sums of chunk 0 have been read)
  commit(std::integral_constant<int, 0>{}, 1, R1, cc, cg); advance(cc); advance(cg);      // (requests dfeats of chunk 2)
  __syncthreads();
  auto iteration = [&](auto st_c, f32x4* RQ, f32x4* RC) __attribute__((always_inline)) {
    constexpr int ST = decltype(st_c)::value;
    issue_x(ci, RQ); advance(ci);
    const f32x16 acc = compute(ST);
    epilogue(ce, acc); advance(ce);
    // behind the awaited dfeats request: the stores of the commit that issued it, this iteration's chunk request, its epilogue
    commit(std::integral_constant<int, (RES ? 8 : 4) + 5 + 4>{}, (ST + 2) % 3, RC, cc, cg); advance(cc); advance(cg);
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
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

bool csn_wx_lnb_takes(const CsnLnBwdArgs& a, int d_inner) {
  return (csn_dev_wx & 9) == 9 && a.C == WX_K && d_inner == WX_K && !a.act16 && !(a.ld & 3) && !(a.n_points & 3) && !(a.eval_stride & 3);
}

int csn_launch_wx_lnb(const CsnWxLnbArgs& a, hipStream_t st) {
  if (a.n_items <= 0 || a.n_points <= 0) return 0;
  if ((long long)a.n_items * ((a.n_points + WX_CH - 1) / WX_CH + 16) * 2 >= (1ll << 31)) return CSN_NOT_TAKEN;
  CsnWxLnbArgs b = a;
  b.ablate = (csn_dev_wx >> 4) & 15;
  const int grid = wx_grid();
  // (always the instance with the mask code: at p = 0 the threshold is 0 and the scale 1 — every element kept, times 1.0.  The
  //  instances without it compile to 90 spilled registers: the scheduler then hoists the whole commit's loads)
  if (a.dz_res) hipLaunchKernelGGL((csn_wx_lnb_kernel<true, true>), dim3(grid), dim3(512), 0, st, b);
  else hipLaunchKernelGGL((csn_wx_lnb_kernel<true, false>), dim3(grid), dim3(512), 0, st, b);
  return (int)hipGetLastError();
}
