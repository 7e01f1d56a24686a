// Fused block-diagonal attention for the CSA layer in the "bf16x3" math mode: every fp32 product is three bf16
// matrix-core products (hi*hi + hi*lo + lo*hi, fp32 accumulate; see gemm_bf16x3.hip for the error model).
//
// Same skeleton, tiling, LDS budget and HBM data flow as attn_f32.hip (read that header first):
//   R      [d][q]   register operand, 16 query points per wave        (fwd: Qs^T   bwd: dO^T)    as bf16 hi/lo fragments
//   tileA  [d][key] 32 keys, k-major: fragments by ds_read_b64_tr_b16  (fwd: K^T    bwd: V^T)     bf16 hi / lo planes
//   tileB  [d][key] 32 keys, key-contiguous: fragments by ds_read_b128 (fwd: V^T    bwd: K^T)     bf16 hi / lo planes
// Matrix instruction: v_mfma_f32_16x16x32_bf16 (A: lane l = A[l & 15][8 (l >> 4) + j], B: lane l = B[8 (l >> 4) + j][l & 15],
// j = 0..7; C: reg r of lane l = C[4 (l >> 4) + r][l & 15]).  The two 16-row score tiles of a 32-key tile take the keys
//   S0 row i  <->  key 8 (i >> 2) + (i & 3),      S1 row i  <->  key 8 (i >> 2) + 4 + (i & 3)
// (the transposing LDS read takes a per-lane address, so any 4-key chunk can feed any row group), hence lane (q, kq)
// ends phase 1 with the 8 CONSECUTIVE keys 8 kq .. 8 kq + 7 of its query: after the hi/lo split that is the B fragment of
// phase 2 as it stands, and the V^T / K^T fragment of a lane is one 16-byte run of the natural [d][key] row.
// K/V inputs, two forms:
//   fp32 [d][point]  — tiles are split into bf16 hi/lo planes while they are staged into LDS;
//   "tile planes"    — written by the projection (csn_project_f32, out_split = 2): per row and 500-point block, 16 tiles of
//                      [hi: 32 keys | lo: 32 keys] bf16 (block pitch 1024, padding zero).  A tile row is then 128
//                      contiguous, 128-byte aligned bytes, and staging is a plain copy — no conversion work per tile.
// LDS images (conflict-free): tileA swaps the two 8-byte chunks of every 16-byte unit on rows with bit 3 ^ bit 0 set (the
// transposing read of a 32-lane group touches rows r and r + 8 together, a staging store rows r and r + 1); tileB XORs the
// 16-byte unit index with (-(row >> 2)) & 3; the hi and lo planes are 64 bytes out of phase so that one staging store (whose
// banks repeat every 128 bytes) hits both without conflict.
// Schedule: LDS fragment reads run 4 steps ahead of their matrix instructions (explicit register ring); a tile is two barrier
// segments and waves 4..7 run one segment behind waves 0..3; the register operand, the delta inputs and the output travel
// as 16-byte rows through an LDS transpose (a wave-level memory instruction costs ~100 cycles whatever its width).
#include "csn_common.h"
#include "csn_kernels.h"

#ifdef CSN_STAMPS
__device__ unsigned long long csn_dbg[2048 * 8 * 4 * 8];
extern "C" __attribute__((visibility("default"))) int csn_debug_read(void* dst, long long bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(csn_dbg), bytes); }
// whole-kernel stamps per wave: entry, tile loop start, tile loop end, exit
__device__ unsigned long long csn_dbg_wg[2048 * 8 * 4];
__device__ unsigned long long csn_dbg_rt[2048 * 8 * 2];     // s_memrealtime (100 MHz) at entry and exit: in-kernel clock
extern "C" __attribute__((visibility("default"))) int csn_debug_read_wg(void* dst, long long bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(csn_dbg_wg), bytes); }
extern "C" __attribute__((visibility("default"))) int csn_debug_read_rt(void* dst, long long bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(csn_dbg_rt), bytes); }
#define WGSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); if ((BWD == (CSN_STAMPS != 0)) && DT == 8 && blockIdx.x >= 4096 && blockIdx.x < 6144 && (threadIdx.x & 63) == 0) { csn_dbg_wg[((blockIdx.x - 4096) * 8 + (threadIdx.x >> 6)) * 4 + i] = __builtin_amdgcn_s_memtime(); if (i == 0 || i == 3) csn_dbg_rt[((blockIdx.x - 4096) * 8 + (threadIdx.x >> 6)) * 2 + (i == 3)] = __builtin_amdgcn_s_memrealtime(); } __builtin_amdgcn_sched_barrier(0); } while (0)
// prologue stamps of the same work-groups: entry, operand block requested, landed (barrier), picked, tiles fetched and committed
__device__ unsigned long long csn_dbg_pro[2048 * 8 * 8];
extern "C" __attribute__((visibility("default"))) int csn_debug_read_pro(void* dst, long long bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(csn_dbg_pro), bytes); }
#define PSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); if ((BWD == (CSN_STAMPS != 0)) && DT == 8 && blockIdx.x >= 4096 && blockIdx.x < 6144 && (threadIdx.x & 63) == 0) csn_dbg_pro[((blockIdx.x - 4096) * 8 + (threadIdx.x >> 6)) * 8 + i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WGSTAMP(i)
#define PSTAMP(i)
#endif

namespace {

constexpr int KT = 32;               // keys per streamed tile
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

using namespace csn_mode;
typedef f32x4m f32x4v;
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
typedef short __attribute__((address_space(3))) lds_s16;
typedef s16x8 __attribute__((address_space(3))) lds_s16x8;

// An LDS address the compiler cannot fold into its constant arithmetic.  The V^T / K^T tile array starts 64 KB into the LDS
// block, beyond the 16-bit immediate of a DS instruction: left to itself the compiler rebuilt the address of every one of the
// 2 x D/16 fragment reads of phase 2 with a vector add (26 per tile at d = 256).  With the stage's base made opaque once per
// phase, every read is that register plus an immediate.
CSN_DEVINL const lds_s16* opaque_lds(const short* p) {
  const lds_s16* q = (const lds_s16*)p;
  asm volatile("" : "+v"(q));
  return q;
}

// acc += a * b on the 16x16x32 matrix instruction: a = ah + al, b = bh + bl as three products (small terms first), or one
template <typename PR>
CSN_DEVINL f32x4v mma16(s16x8 ah, s16x8 al, s16x8 bh, s16x8 bl, f32x4v c) {
  if constexpr (PR::NT == 3) {
    c = mfma16<PR::HALF>(al, bh, c);
    c = mfma16<PR::HALF>(ah, bl, c);
  }
  return mfma16<PR::HALF>(ah, bh, c);
}

// PR = csn_mode::Bf16x3 (math mode 1: hi / lo planes, three products), Bf16 / F16 (modes 2 / 3: one plane, one product;
// tile-plane K/V only, F16 forward only).  The single-product modes drop every "lo" object of this file: the LDS planes, the
// fragment reads, the conversions, two of the three matrix instructions, and half of the bytes of every tile plane.
// RC (backward, tile-plane K / V only): the scores are RECOMPUTED — S = Qs K^T from a second register operand (Qs^T of the
// query slot) and a third LDS image (the K tile in the k-major form of tileA) — instead of being read back from the forward's
// saved copy: no score loads, and with sc_tiles == 0 no P / dS stores either (the key-stationary dK / dV kernel of
// attn_dkv.hip recomputes them for itself).  One more matrix product per tile; LDS holds three images per stage.
// Waves per SIMD (the second launch bound).  Two everywhere — 256 registers a wave — except the narrow instances, whose tile
// images leave room for a second work-group on the CU: one plane up to d = 96, two planes up to d = 64.  There the bound is
// four (128 registers), which costs the recomputing dQ instances 13-18 spilled registers and still pays: a second work-group
// hides what a latency-bound loop cannot (config-5 geometry, bf16: dQ 3.53 -> 2.69 ms; bf16x3 at d = 64: 2.43 -> 1.75 ms).
// Measured and NOT taken (scripts/dev/ab_attn.sh): one plane at d = 128 (+-0), two planes at d = 96 / 128 with the default
// ring depth (38-82 spilled registers inside the loop: forward 3.8 -> 6.6 ms; the d = 96 forward with a ring of two: below).
#ifndef CSN_LB_NARROW
#define CSN_LB_NARROW 4
#endif
#ifndef CSN_LB_X3FWD96
#define CSN_LB_X3FWD96 1
#endif
// (two planes, d = 96, FORWARD: with a fragment ring two deep instead of four it fits 128 registers with 4 spills)
constexpr bool csn_attn_x3fwd96(int npl, int dt, bool bwd) { return CSN_LB_X3FWD96 && npl == 2 && dt == 3 && !bwd; }
constexpr int csn_attn_waves(int npl, int dt, bool bwd) {
  return ((npl == 1 ? dt <= 3 : dt <= 2) || csn_attn_x3fwd96(npl, dt, bwd)) ? CSN_LB_NARROW : 2;
}
template <typename PR, int DT, bool BWD, bool KVP, bool RC = false>
__global__ __launch_bounds__(512, csn_attn_waves(PR::NPL, DT, BWD)) void csn_attn_bf16x3_kernel(CsnAttnArgs p) {
  static_assert(PR::NT == 3 || KVP, "single-product modes take K / V as tile planes");
  static_assert(!RC || (BWD && KVP), "score recomputation: backward kernel on tile-plane K / V");
  constexpr int NPL = PR::NPL;                          // planes: hi (+ lo)
  constexpr int D = 32 * DT;
  constexpr int UPR = KVP ? 4 * NPL : 8;                // 16-byte pieces per tile row (tile planes: 4 per plane; fp32: 8)
  constexpr int RPP = 512 / UPR;                        // tile rows staged per pass of the 512 threads
  constexpr int PIECES = D * UPR;                       // 16-byte pieces per streamed tile
  constexpr int NP_T = (PIECES + 511) / 512;            // pieces per thread per tile
  // plane pitch: 64 bytes of phase between hi and lo.  LDS STORES bank on 32 dwords (128 bytes): a staging instruction of the
  // tile-plane path writes the hi and the lo unit of a row together (lanes u and u + 4), and with the planes a multiple of
  // 128 bytes apart the two would land on the same banks (measured: SQ_LDS_BANK_CONFLICT = 15 % of the LDS cycles of both
  // attention kernels, all of it these stores).  CSN_LDS_V=0 rebuilds the old image (128 bytes of phase) for A/B timing.
#ifndef CSN_LDS_V
#define CSN_LDS_V 1
#endif
// (1: the forward asks for the first K / V tile before it stages its operand block, 2: the backward too.  Measured: forward
//  5.17 / 5.16 ms against 5.17 / 5.14 without, backward 7.41 / 7.33 against 7.14 / 7.06 — 53 spilled registers instead of 19;
//  profiles/r4u_attention_prologue_and_priority.txt.  Off.)
#ifndef CSN_PREFETCH_TILE0
#define CSN_PREFETCH_TILE0 0
#endif
  constexpr int PLANE = D * KT + (CSN_LDS_V ? 32 : 64);
  // [A | B][stage][plane hi/lo][row][32 keys] — one array, so that the prologue / epilogue can use all of it as a
  // [D rows][128 queries] fp32 staging block for 16-byte global accesses (which sets the size in the one-plane modes)
  // EARLY: the narrow instances (and every one-plane instance) request the tiles a whole segment earlier, into a second
  // register set — their matrix phases are too short to cover the HBM latency of a request made one phase before its use
#ifndef CSN_EARLY
#define CSN_EARLY 1
#endif
#ifndef CSN_EARLY_ALL
#define CSN_EARLY_ALL 0
#endif
  // Round 5: the two-plane d = 256 FORWARD too — its second register set costs no spills there (255 registers) — forward alone
  // 5.29 -> 5.22 ms, config-3 step 25.94 -> 25.76 ms over four alternations (profiles/r5q_forward_early_requests.txt); the
  // backward at this width does not gain (-DCSN_EARLY_ALL=1: 7.00 -> 7.10 ms)
#ifndef CSN_EARLY_FWD8
#define CSN_EARLY_FWD8 1
#endif
  constexpr bool EARLY = CSN_EARLY && (NPL == 1 || DT <= 4 || CSN_EARLY_ALL || (CSN_EARLY_FWD8 && !BWD && !RC));
  // LDS images: A x 2 stages, B x 2 (RC && EARLY: x 3 — the key-contiguous K image is then committed in segment 1 too, which
  // needs a third stage), RC: + C x 2 (the K tile in tileA's form)
  constexpr int NB_ST = (RC && EARLY) ? 3 : 2;
  constexpr bool C_ALIASED = CSN_RC_ALIAS && RC && NPL == 2 && DT > 4;        // timing experiment: image C laid over image B
  constexpr int TILE_EL = (2 + NB_ST + (RC && !C_ALIASED ? 2 : 0)) * NPL * PLANE, STAGE_EL = D * 128 * 2;
  static_assert(2 * (TILE_EL > STAGE_EL ? TILE_EL : STAGE_EL) <= 160 * 1024, "LDS budget of one CU");
  __shared__ __attribute__((aligned(16))) short tiles[TILE_EL > STAGE_EL ? TILE_EL : STAGE_EL];
  auto tileA = [&](int st, int pl) -> short* { return tiles + (st * NPL + pl) * PLANE; };
  auto tileB = [&](int st, int pl) -> short* { return tiles + ((2 + st) * NPL + pl) * PLANE; };
  auto tileC = [&](int st, int pl) -> short* { return tiles + ((2 + (C_ALIASED ? 0 : NB_ST) + st) * NPL + pl) * PLANE; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 15, kq = lane >> 4;
  WGSTAMP(0);
  // XCD-aware work-group order.  A unit = one (evaluation, head, block); its QT query tiles all stream the same K/V
  // block, so they should share one XCD's L2.  Work-groups are dealt round-robin over the 8 XCDs, hence the QT tiles of
  // unit u get ids 8 * (QT * (u / 8) + qt) + (u % 8): same residue mod 8 (same XCD), adjacent in dispatch order.
  // (Placement only changes speed: every tile is self-contained.)
  const int Tq = p.Tq > 0 ? p.Tq : p.T;                       // queries per block (T: keys per block)
  const int QT = (Tq + 127) / 128;
  const int Y = p.n_blocks * p.H;
  const int L = blockIdx.x, slot = L & 7, jj = L >> 3;
  const int qt = jj % QT, u = (jj / QT) * 8 + slot;
  if (u >= Y * p.E) return;
  // items of this work-group.  Plain launches: the one evaluation eval_ids[u / Y].  Grouped backward (grp_off != null): the
  // launch index counts GROUPS of evaluations that share the output slot (the query shape's dQ collects K+2 evaluations);
  // the group's evaluations eval_ids[grp_off[g] .. grp_off[g+1]) run one after the other into the same OUT accumulators,
  // which leave once — no read-modify-write passes over the gradient maps
  const int zz = u / Y;
  // Grouped forward (round 5, csn_block_attn_fwd_grouped_f32): the group's evaluations share the QUERY slot — the register operand
  // Qs is staged once and stays in registers while the group's evaluations run one after the other (own K / V, own scores,
  // own output: the epilogue is inside the loop)
  const int it0 = p.grp_off ? p.grp_off[zz] : zz;
  const int it1 = p.grp_off ? p.grp_off[zz + 1] : it0 + 1;
  const int e0 = p.eval_ids ? p.eval_ids[it0] : it0;
  const int hd = (u % Y) % p.H, blk = (u % Y) / p.H;
  // ragged batches: this evaluation's own query / key counts (Tq, p.T stay the maxima that lay out the buffers)
  const bool short_blk = p.T_last > 0 && blk == p.n_blocks - 1;     // the row ends inside the last block
  const int Tq_e = p.tq_arr ? p.tq_arr[e0] : (short_blk ? p.T_last : Tq);
  if (qt * 128 >= Tq_e) return;                                // a query tile beyond a short evaluation (whole work-group)
  const int T = p.t_arr ? p.t_arr[e0] : (short_blk ? p.T_last : p.T), Tp = p.Tp, ld = p.ld, ldk = p.ld_kv > 0 ? p.ld_kv : p.ld;
  const bool ragged = (T & 3) != 0;                            // keys of the last 4-key group are masked one by one
  const int qrow = qt * 128 + wave * 16 + lq;                  // query index inside the block
  const bool q_ok = qrow < Tq_e;
  const long long head_off = (long long)hd * D * ld + (long long)blk * Tq;
  const long long win = ((long long)(D - 1) * ld + Tq) * 4;    // bytes spanned by a [D][Tq] fp32 window of pitch ld
  long long os = p.out_index ? p.out_index[e0] : e0;
  // 16-bit activation maps (single-product modes; r_fmt / ctx_fmt / q2_fmt / out_fmt: 0 fp32, 1 bf16, 2 fp16): a [D][Tq] window
  // of a map in format f starts `el` elements into it
  auto map_rsrc = [&](const void* base, long long el, int f) {
    return f ? csn_make_rsrc(reinterpret_cast<const short*>(base) + el, win / 2) : csn_make_rsrc(reinterpret_cast<const float*>(base) + el, win);
  };
  long long stat_off0 = ((long long)e0 * p.H + hd) * ((long long)p.n_blocks * Tq) + (long long)blk * Tq;
  float* xbuf = reinterpret_cast<float*>(tiles);
  constexpr int CH_T = D / 16;                                     // 16-byte chunks per thread: D rows x 32 chunks / 512
  const int cc = tid & 31, crow = tid >> 5;                        // chunk column, first row of this thread (rows + 16 t)
  const unsigned c_off = (qt * 128 + 4 * cc) < Tq_e ? (unsigned)(crow * ld + qt * 128 + 4 * cc) * 4u : CSN_OOB;
  const int col = 16 * wave + lq;                                  // this lane's query column inside the block of 128
  const bool late = __builtin_amdgcn_readfirstlane(wave) >= 4;
  f32x4v O[D / 16];
#pragma unroll
  for (int c = 0; c < D / 16; ++c) O[c] = f32x4v{0.f, 0.f, 0.f, 0.f};
  s16x8 Rh[D / 32], Rl[D / 32];                          // the register operand (forward groups: staged by the first item)
  s16x8 Qh[RC ? D / 32 : 1], Ql[RC ? D / 32 : 1];        // score recomputation: Qs^T of the query slot, a second register operand
  long long q2s_staged = -1;
  float m_run = -INFINITY, m2_run = -INFINITY, l_run = 0.f;   // forward: running max / partial sum of this lane's key quarter

  // ---- epilogue (backward: once, behind the group's last item; forward: behind every item) ----------------------------
  auto write_out = [&]() __attribute__((always_inline)) {
    float inv = 1.f;
    if (!BWD) {
      float l_tot = l_run + __shfl_xor(l_run, 16, 64);
      l_tot += __shfl_xor(l_tot, 32, 64);
      inv = 1.f / l_tot;
      if (q_ok && kq == 0 && p.lse) p.lse[stat_off0 + qrow] = m2_run * LN2 + logf(l_tot);
    }
    // OUT leaves through the same [D][128] LDS block as 16-byte rows (chunk c of row r at c ^ 4 ((r >> 2) & 1): the lane
    // quarters write rows 4 apart); when several evaluations share the output slot, the previous partial sums are fetched
    // first — one batch of 16-byte loads in flight — then added and stored
  #pragma unroll
    for (int c = 0; c < D / 16; ++c)
  #pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * c + 4 * kq + r;
        xbuf[row * 128 + ((((col >> 2) ^ (4 * ((row >> 2) & 1))) << 2) | (col & 3))] = O[c][r] * inv;
      }
    __syncthreads();
    {
      f32x4 ch[CH_T];
  #pragma unroll
      for (int t = 0; t < CH_T; ++t) {
        const int row = crow + 16 * t;
        ch[t] = *reinterpret_cast<const f32x4*>(&xbuf[row * 128 + ((cc ^ (4 * ((row >> 2) & 1))) << 2)]);
      }
      const csn_rsrc_t Or = map_rsrc(p.out, os * p.out_eval_stride + head_off, p.out_fmt);
      if (NPL == 1 && p.out_fmt) {                                   // a 16-bit map (written once: the launcher refuses accumulate)
  #pragma unroll
        for (int t = 0; t < CH_T; ++t) {
          const s16x4 v = p.out_fmt == 2 ? to16x4<true>(ch[t]) : to16x4<false>(ch[t]);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), Or, c_off == CSN_OOB ? CSN_OOB : c_off >> 1, (unsigned)(16 * t * ld) * 2u, 0);
        }
      } else {
        if (p.accumulate) {
          f32x4 prev[CH_T];
  #pragma unroll
          for (int t = 0; t < CH_T; ++t) prev[t] = csn_bload4(Or, c_off, (unsigned)(16 * t * ld) * 4u);
  #pragma unroll
          for (int t = 0; t < CH_T; ++t) ch[t] += prev[t];
        }
  #pragma unroll
        for (int t = 0; t < CH_T; ++t) csn_bstore4(ch[t], Or, c_off, (unsigned)(16 * t * ld) * 4u);
      }
    }
  };

  for (int it = it0; it < it1; ++it) {
  const int e = p.eval_ids ? p.eval_ids[it] : it;
  const long long qs = p.q_index ? p.q_index[e] : e;
  const long long ks = p.kv_index ? p.kv_index[e] : e;
  const long long head_off_kv = (long long)hd * D * ldk + (long long)blk * p.T;   // (p.T: the layout; T may be a short last block)
  const long long win_kv = ((long long)(D - 1) * ldk + (T + 3) / 4 * 4) * 4;
  const csn_rsrc_t Rr = map_rsrc(p.q, qs * p.q_shape_stride + head_off, p.r_fmt);
  // tile planes: 16-bit elements, row pitch kv_ld = n_blocks * 512 NPL, this block's 16 tiles start at blk * 512 NPL
  const int kld = p.kv_ld;
  const long long kv_off = KVP ? ks * p.kv_shape_stride + (long long)hd * D * kld + (long long)blk * (512 * NPL) : 0;
  const long long kv_win = ((long long)(D - 1) * kld + 512 * NPL) * 2;
  const short* kpl = reinterpret_cast<const short*>(BWD ? p.v : p.k);
  const short* vpl = reinterpret_cast<const short*>(BWD ? p.k : p.v);
  const csn_rsrc_t Ar = KVP ? csn_make_rsrc(kpl + kv_off, kv_win)
                            : csn_make_rsrc((BWD ? p.v : p.k) + ks * p.kv_shape_stride + head_off_kv, win_kv);
  const csn_rsrc_t Br = KVP ? csn_make_rsrc(vpl + kv_off, kv_win)
                            : csn_make_rsrc((BWD ? p.k : p.v) + ks * p.kv_shape_stride + head_off_kv, win_kv);
  const long long stat_off = ((long long)e * p.H + hd) * ((long long)p.n_blocks * Tq) + (long long)blk * Tq;
  const long long sc_off = (((long long)e * p.H + hd) * p.n_blocks + blk) * ((long long)Tq * Tp);
  const bool have_scores = p.scores != nullptr;
  const csn_rsrc_t Sr = csn_make_rsrc(have_scores ? p.scores + sc_off : nullptr, have_scores ? (long long)Tq * Tp * 4 : 0);
  const csn_rsrc_t dSr = csn_make_rsrc(BWD ? p.dscores + sc_off : nullptr, BWD ? (long long)Tq * Tp * 4 : 0);

  // per-lane byte offsets (scalar offsets handed to the buffer instructions must be wave-uniform, so
  // everything that depends on the lane lives here); lanes of query rows beyond the block are switched off

  // ---- streamed tiles: global -> registers -> LDS (swizzled) ---------------------------------------
  // piece idx = tid + 512 i  ->  row tid / UPR + RPP i.  fp32 input: keys 4 c .. 4 c + 3, c = tid % 8.  Tile planes: 16-byte
  // unit tid % UPR of the row's 64 NPL bytes (units 0..3: hi plane keys 8 u .. 8 u + 7, units 4..7: lo plane).  Everything that
  // depends on i is wave-uniform (scalar offset of the load, immediate offset of the LDS store): rows 64 apart share the swizzles.
  const int t_c = tid & (UPR - 1), t_row = tid / UPR;
  constexpr int SWB = (KVP && NPL == 1) ? 1 : 0;        // one plane: a 16-lane store group covers rows r .. r + 3 — swap on bit 1
  // tileA chunk swap: rows r and r + 8 are read together by the transposing read, and the 8-byte stores of a 16-lane group
  // cover rows r, r + 1 of both planes — so the swap bit is (r >> 3) ^ r: both pairs then sit on complementary banks
  const int t_sw = (CSN_LDS_V ? ((t_row >> 3) ^ (t_row >> SWB)) : (t_row >> 3)) & 1, t_swz = (-((t_row >> 2) & 3)) & 3;
  const unsigned t_off = KVP ? (unsigned)(t_row * kld * 2 + t_c * 16) : (unsigned)(t_row * ldk + 4 * t_c) * 4u;
  // only the last piece can fall beyond the tile — and not even that one when the pieces fill the passes (compile-time: the
  // guards around the last piece's stores fold away)
  const bool t_last_ok = (PIECES % 512 == 0) || (tid + 512 * (NP_T - 1) < PIECES);
  // fp32: 8-byte chunk c -> tileA chunk c ^ sw;  tileB unit (c >> 1) ^ swz, half c & 1
  // planes: unit u = c & 3 of plane c >> 2 -> tileA chunks (2 u) ^ sw and (2 u + 1) ^ sw;  tileB unit u ^ swz
  const int t_u = t_c & 3, t_pl = t_c >> 2;
  const int a_dst = KVP ? t_pl * PLANE + t_row * KT + 8 * t_u : t_row * KT + 4 * (t_c ^ t_sw);
  const int b_dst = KVP ? t_pl * PLANE + t_row * KT + 8 * (t_u ^ t_swz) : t_row * KT + 8 * ((t_c >> 1) ^ t_swz) + 4 * (t_c & 1);
  f32x4 g[NP_T];
  f32x4 g2[(RC || EARLY) ? NP_T : 1];                   // RC: the K tile's pieces (committed to two images) beside the V tile's; EARLY: tileB's
  auto fetch_to = [&](const csn_rsrc_t& rs, int kt, f32x4* g) {
    if (KVP) {
      // keys beyond the block end are zero in the planes; units that lie entirely beyond it are not fetched at all
      const unsigned off = (kt * KT + 8 * t_u) < T ? t_off : CSN_OOB;
#pragma unroll
      for (int i = 0; i < NP_T; ++i)
        g[i] = csn_bload4(rs, (i == NP_T - 1 && !t_last_ok) ? CSN_OOB : off, (unsigned)(kt * (64 * NPL) + RPP * i * kld * 2));
    } else {
      const int k0 = kt * KT;
      // T % 4 == 0: a 16-byte piece is all in or all out; pieces past the block end are switched off
      const unsigned off = (k0 + 4 * t_c) < T ? t_off : CSN_OOB;
#pragma unroll
      for (int i = 0; i < NP_T; ++i)
        g[i] = csn_bload4(rs, (i == NP_T - 1 && !t_last_ok) ? CSN_OOB : off, (unsigned)(k0 + 64 * i * ldk) * 4u);
    }
  };
  // The first tile's pieces are asked for BEFORE the operand block is staged (the stamps of the d = 256 forward: 21 k of a
  // work-group's 135 k cycles were this prologue, most of it three memory round trips in a row — operand block, K tile, V
  // tile — with nothing else in flight; profiles/r4u_attention_forward_stamps.txt): they travel beside the operand block and
  // wait in registers until the staging block has been read
  constexpr bool PRE0 = CSN_PREFETCH_TILE0 && !RC && !EARLY && (!BWD || CSN_PREFETCH_TILE0 > 1);      // (backward: 19 -> 53 spilled registers)
  f32x4 gp[PRE0 ? NP_T : 1];
  if constexpr (PRE0) { fetch_to(Ar, 0, g); fetch_to(Br, 0, gp); }

  // ---- register-resident operand R[d][q]: lane (q, kq) keeps rows d = 32 s + 8 kq + j as bf16 hi / lo ------
  // A wave-level memory instruction costs ~100 cycles of issue whatever its width, and the register layout would need
  // 64 four-byte loads per lane (128 in the backward, which also reads O for delta).  So the work-group fetches its
  // [D][128 queries] block with 16-byte loads (D/16 per thread) into LDS — the tile buffers are still idle — and every
  // lane picks its values from there.  Chunk c of row r sits at c ^ 4 ((r >> 3) & 1): rows 8 apart (lane quarters kq, kq+1)
  // use different banks.  Backward: delta_q = sum_d dO[d][q] O[d][q] (the softmax-backward row constant) from a second
  // round with O.
  auto stage_in = [&](const csn_rsrc_t& rs, int fmt) {
    f32x4 ch[CH_T];
    if (NPL == 1 && fmt) {                                         // a 16-bit map: half the bytes, widened on the way into LDS
      u32x2 c2[CH_T];
#pragma unroll
      for (int t = 0; t < CH_T; ++t) c2[t] = csn_bload2(rs, c_off == CSN_OOB ? CSN_OOB : c_off >> 1, (unsigned)(16 * t * ld) * 2u);
#pragma unroll
      for (int t = 0; t < CH_T; ++t) ch[t] = act16_to_f32(__builtin_bit_cast(s16x4, c2[t]), fmt);
    } else {
#pragma unroll
      for (int t = 0; t < CH_T; ++t) ch[t] = csn_bload4(rs, c_off, (unsigned)(16 * t * ld) * 4u);
    }
#pragma unroll
    for (int t = 0; t < CH_T; ++t) {
      const int row = crow + 16 * t;
      *reinterpret_cast<f32x4*>(&xbuf[row * 128 + ((cc ^ (4 * ((row >> 3) & 1))) << 2)]) = ch[t];
    }
  };
  auto pick = [&](int row) { return xbuf[row * 128 + ((((col >> 2) ^ (4 * ((row >> 3) & 1))) << 2) | (col & 3))]; };
  PSTAMP(0);
  float rv[D / 4];
  if (BWD || it == it0) {                                          // (forward group: the operand of the first item serves them all)
    stage_in(Rr, p.r_fmt);
    PSTAMP(1);
    __syncthreads();
    PSTAMP(2);
#pragma unroll
    for (int s = 0; s < D / 32; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = pick(32 * s + 8 * kq + j);
        rv[8 * s + j] = v;
        Rh[s][j] = to16<PR::HALF>(v);
        Rl[s][j] = PR::NT == 3 ? to16<PR::HALF>(v - from16<PR::HALF>(Rh[s][j])) : Rh[s][j];
      }
  }
  if (!BWD) {                                                      // forward: every item has its own output and statistics
    os = p.out_index ? p.out_index[e] : e;
    stat_off0 = stat_off;
#pragma unroll
    for (int c = 0; c < D / 16; ++c) O[c] = f32x4v{0.f, 0.f, 0.f, 0.f};
    m_run = -INFINITY; m2_run = -INFINITY; l_run = 0.f;
  }
  PSTAMP(3);
  float delta_q = 0.f;
  if (BWD) {
    const csn_rsrc_t Xr = map_rsrc(p.ctx, qs * p.q_shape_stride + head_off, p.ctx_fmt);
    __syncthreads();
    stage_in(Xr, p.ctx_fmt);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < D / 32; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) delta_q = fmaf(rv[8 * s + j], pick(32 * s + 8 * kq + j), delta_q);
    delta_q += __shfl_xor(delta_q, 16, 64);
    delta_q += __shfl_xor(delta_q, 32, 64);
  }
  // score recomputation: the pre-scaled queries Qs^T of this evaluation's query slot, a second register operand
  // (a group's evaluations share their query slot in the training step: the operand is staged when the slot changes — once)
  if constexpr (RC) {
    const long long q2s = p.q2_index ? p.q2_index[e] : e;
    if (it == it0 || q2s != q2s_staged) {                 // (measured: -0.05 ms of the config-3 bf16 step, within the spread; fewer bytes)
      q2s_staged = q2s;
      const csn_rsrc_t Qr = map_rsrc(p.q2, q2s * p.q2_shape_stride + head_off, p.q2_fmt);
      __syncthreads();
      stage_in(Qr, p.q2_fmt);
      __syncthreads();
#pragma unroll
      for (int s = 0; s < D / 32; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = pick(32 * s + 8 * kq + j);
          Qh[s][j] = to16<PR::HALF>(v);
          Ql[s][j] = PR::NT == 3 ? to16<PR::HALF>(v - from16<PR::HALF>(Qh[s][j])) : Qh[s][j];
        }
    }
  }
  __syncthreads();                                                 // the staging block becomes the tile buffers

  // attention-probability dropout (csa_models.py:141): P_drop = mask * P / (1 - p); element index = position in `scores`
  const bool drop = p.dropout_p > 0.f;
  const unsigned thr16 = csn_drop_threshold16(p.dropout_p);
  const float keep_scale = drop ? 1.f / (1.f - p.dropout_p) : 1.f;
  const unsigned salt = csn_block_salt((unsigned long long)(((long long)e * p.H + hd) * p.n_blocks + blk), p.seed);
  const int mp = Tq > Tp ? Tq : Tp;                               // mask pitch: pair index = key pair * mp + query stays unique when n_queries > score_pitch
  const unsigned pw_base = (unsigned)(4 * kq * mp + qrow);       // pair index of this lane's keys 8 kq, 8 kq + 1

  // exponentials run on the hardware exp2: exp(s - m) = exp2(s * log2(e) - m2) with m2 = fl(m * log2(e)), the same m2 for
  // every key of a query, so its rounding cancels in the normalisation; lse is rebuilt from m2 (= m2 ln 2 + ln l)
  float lse2_q = 0.f;                          // backward: per-query constant
  if (BWD) {
    lse2_q = q_ok ? p.lse[stat_off + qrow] * LOG2E : 0.f;
    if (q_ok && kq == 0 && p.delta) p.delta[stat_off + qrow] = delta_q;
  }
  // score positions of this lane: the scores of a block are stored [query][key] (pitch Tp), so the 8 consecutive keys
  // kt*32 + 8 kq .. + 7 of this lane's query are two 16-byte runs (half j = keys + 4 j .. + 3: all in or all out, T % 4 == 0)
  // (tile-major storage: tile kt of the block is [query][32 keys], Tq * 128 bytes per tile)
  const bool tile_major = p.sc_layout != 0;
  const unsigned s_base = tile_major ? (unsigned)(qrow * KT + 8 * kq) * 4u : (unsigned)(qrow * Tp + 8 * kq) * 4u;
  const unsigned s_tile = tile_major ? (unsigned)Tq * (KT * 4u) : KT * 4u;      // bytes from one key tile to the next

  // backward of an fp16 forward (math mode 3): the K / V tile planes hold fp16 bits and the backward's products are bf16 —
  // every piece is converted once, in registers, on its way into LDS (instead of projecting K and V a second time)
  constexpr bool CAN_CVT = BWD && KVP && PR::NPL == 1 && !PR::HALF;
  const bool kv_f16 = CAN_CVT && p.kv_f16;
  auto cvt_pieces = [&](f32x4* g) {
    if constexpr (CAN_CVT) {
      if (kv_f16) {
#pragma unroll
        for (int i = 0; i < NP_T; ++i) {
          s16x8 v = __builtin_bit_cast(s16x8, g[i]);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = to16<false>(from16<true>(v[j]));
          g[i] = __builtin_bit_cast(f32x4, v);
        }
      }
    }
  };
  auto commitA_to = [&](short* img, short* img_lo, f32x4* g, bool cvt = true) {       // img / img_lo: hi / lo plane of a k-major image
    if (cvt) cvt_pieces(g);
#pragma unroll
    for (int i = 0; i < NP_T; ++i)
      if (i < NP_T - 1 || t_last_ok) {
        if (KVP) {
          short* base = img + a_dst + RPP * KT * i;
          const s16x8 v = __builtin_bit_cast(s16x8, g[i]);
          *reinterpret_cast<s16x4*>(base + 4 * t_sw) = s16x4{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<s16x4*>(base + 4 * (t_sw ^ 1)) = s16x4{v[4], v[5], v[6], v[7]};
        } else {
          s16x4 hi, lo;
          split4<PR>(g[i], hi, lo);
          *reinterpret_cast<s16x4*>(img + a_dst + RPP * KT * i) = hi;
          *reinterpret_cast<s16x4*>(img_lo + a_dst + RPP * KT * i) = lo;
        }
      }
  };
  auto commitB_from = [&](int st, f32x4* g, bool cvt = true) {
    if (cvt) cvt_pieces(g);
#pragma unroll
    for (int i = 0; i < NP_T; ++i)
      if (i < NP_T - 1 || t_last_ok) {
        if (KVP) {
          *reinterpret_cast<f32x4*>(tileB(st, 0) + b_dst + RPP * KT * i) = g[i];
        } else {
          s16x4 hi, lo;
          split4<PR>(g[i], hi, lo);
          *reinterpret_cast<s16x4*>(tileB(st, 0) + b_dst + RPP * KT * i) = hi;
          *reinterpret_cast<s16x4*>(tileB(st, NPL - 1) + b_dst + RPP * KT * i) = lo;
        }
      }
  };
  auto fetch = [&](const csn_rsrc_t& rs, int kt) { fetch_to(rs, kt, g); };
  auto commitA = [&](int st) { commitA_to(tileA(st, 0), tileA(st, NPL - 1), g); };
  auto commitB = [&](int st) { commitB_from(st, g); };

  // fragment read positions (lane constants).  tileA, transposing read: inside a 16-lane group lane 4 q' + p addresses
  // row 8 kq + q' and the 4-key chunk that feeds score rows 4 p .. 4 p + 3: keys 8 p .. 8 p + 3 for S0, 8 p + 4 .. for S1
  // (chunks swapped inside their 16-byte unit when (row >> 3) & 1 = kq & 1 is set)
  const int tr_row = 8 * kq + (lq >> 2);
  const int tr_sw = (CSN_LDS_V ? (kq ^ (lq >> (2 + SWB))) : kq) & 1;      // = the store side's swap bit of row tr_row (and of tr_row + 4)
  const int a_pos0 = tr_row * KT + 8 * (lq & 3) + 4 * tr_sw, a_pos1 = tr_row * KT + 8 * (lq & 3) + 4 * (tr_sw ^ 1);
  // tileB: row lq of the 16-channel tile, 16-byte unit kq ^ ((-(lq >> 2)) & 3): keys 8 kq .. 8 kq + 7
  const int b_pos = lq * KT + 8 * (kq ^ ((-((lq >> 2) & 3)) & 3));

  const int nkt = (T + KT - 1) / KT;

  // ---- the three phases of one key tile -------------------------------------------------------------
  unsigned s_voff[2];
  bool j_ok[2];
  float sv[8];
  f32x4v S0, S1;
  s16x8 ph, pl;                                         // T1 as a 32-key B fragment, split into hi / lo
  auto score_pos = [&](int kt) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      j_ok[j] = kt * KT + 8 * kq + 4 * j < T;
      s_voff[j] = (q_ok && j_ok[j]) ? s_base + (unsigned)kt * s_tile + (unsigned)(16 * j) : CSN_OOB;
    }
  };
  auto load_sv = [&](int kt) {                          // backward: request the saved scores of tile kt early
    if (BWD && !RC) {
      score_pos(kt);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x4 v = csn_bload4_stream(Sr, s_voff[j]);
        sv[4 * j] = v[0]; sv[4 * j + 1] = v[1]; sv[4 * j + 2] = v[2]; sv[4 * j + 3] = v[3];
      }
    }
  };
  // phase 1: T1[key][q] = sum_d tileA[d][key] R[d][q]
#ifndef CSN_PD
#define CSN_PD 4
#endif
  // LDS fragment reads run CSN_PD steps ahead of the matrix instructions that consume them (explicit register ring):
  // with two waves per SIMD nothing else hides the ~150-cycle LDS latency, and a step is only 48 matrix-pipe cycles.
  constexpr int PD = csn_attn_x3fwd96(NPL, DT, BWD) ? 2 : CSN_PD;
  f32x4v Z0, Z1;                                        // RC: the recomputed scores of this lane's 8 keys (S0 / S1 hold dP)
  auto phase1_on = [&](const short* __restrict__ tAh, const short* __restrict__ tAl, const s16x8* Rh, const s16x8* Rl,
                       f32x4v& S0, f32x4v& S1) {
    S0 = f32x4v{0.f, 0.f, 0.f, 0.f};
    S1 = f32x4v{0.f, 0.f, 0.f, 0.f};
    constexpr int NH = 2 * (D / 32);                    // half steps: (s, 16-key tile t)
    s16x8 ah[PD], al[PD];
    auto rd = [&](int h, s16x8& fh, s16x8& fl) {
      const int o = 32 * (h >> 1) * KT + ((h & 1) ? a_pos1 : a_pos0);
      fh = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tAh + o)),
                 __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tAh + o + 4 * KT)));
      if constexpr (NPL == 2)
        fl = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tAl + o)),
                   __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tAl + o + 4 * KT)));
      else fl = fh;
    };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < PD && h < NH; ++h) rd(h, ah[h], al[h]);
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * NPL * (PD < NH ? PD : NH), 0);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int r = h % PD, sidx = h >> 1;
      if (h & 1) S1 = mma16<PR>(ah[r], al[r], Rh[sidx], Rl[sidx], S1);
      else S0 = mma16<PR>(ah[r], al[r], Rh[sidx], Rl[sidx], S0);
      if (h + PD < NH) rd(h + PD, ah[r], al[r]);
      __builtin_amdgcn_sched_group_barrier(0x008, PR::NT, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * NPL, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto phase1 = [&](int st) {
    phase1_on(tileA(st, 0), tileA(st, NPL - 1), Rh, Rl, S0, S1);
    if constexpr (RC) phase1_on(tileC(st, 0), tileC(st, NPL - 1), Qh, Ql, Z0, Z1);   // S = Qs K^T, as the forward formed it
  };
  // pointwise: softmax / dropout (forward), dS (backward); leaves T1 in (ph, pl)
  auto pointwise = [&](int kt) {
    score_pos(kt);
    // keys beyond the block end exist only in the LAST tile: the masking below is one wave-uniform branch per tile and
    // selects inside it (the per-element form compiled to sixteen exec-mask branches in every tile's vector segment).
    // nv = valid keys among this lane's 8 (queries beyond the block need no mask: their rows are zeros and are never stored)
    const bool last_tile = kt == nkt - 1;
    const int nv = T - (kt * KT + 8 * kq);
    float t1[8] = {S0[0], S0[1], S0[2], S0[3], S1[0], S1[1], S1[2], S1[3]};
    // keep decisions of this lane's 8 elements: one hash per key pair (csn_common.h)
    bool keep[8];
    if (drop) {
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const unsigned h = csn_pair_hash(pw_base + (unsigned)((kt * (KT / 2) + w) * mp), salt);
        keep[2 * w] = (h & 0xffffu) >= thr16;
        keep[2 * w + 1] = (h >> 16) >= thr16;
      }
    }
    if (!BWD) {
      if (last_tile) {
#pragma unroll
        for (int r = 0; r < 8; ++r) t1[r] = r < nv ? t1[r] : -INFINITY;
      }
      float mx = fmaxf(fmaxf(fmaxf(t1[0], t1[1]), fmaxf(t1[2], t1[3])), fmaxf(fmaxf(t1[4], t1[5]), fmaxf(t1[6], t1[7])));
      if (have_scores) {                                       // (not kept: inference, or a backward that recomputes them)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          csn_bstore4_stream(f32x4{t1[4 * j], t1[4 * j + 1], t1[4 * j + 2], t1[4 * j + 3]}, Sr, s_voff[j]);
      }
      // lazy rescale: only when some query's running maximum would grow by more than the threshold.  The four lanes
      // of a query share m_run, so the cross-lane maximum is only needed inside the (rare) branch.
      if (__any(mx > m_run + p.rescale_threshold)) {
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m2_new = m_new * LOG2E;
        const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m2_run - m2_new);
#pragma unroll
        for (int c = 0; c < D / 16; ++c) O[c] *= alpha;
        l_run *= alpha;
        m_run = m_new;
        m2_run = m2_new;
      }
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        t1[r] = __builtin_amdgcn_exp2f(fmaf(t1[r], LOG2E, -m2_run));     // masked keys: exp2(-inf) = 0
        ps += t1[r];
      }
      l_run += ps;                                   // the softmax denominator sees every key, dropped or not
      if (drop) {
#pragma unroll
        for (int r = 0; r < 8; ++r) t1[r] = keep[r] ? t1[r] * keep_scale : 0.f;
      }
    } else {
      if constexpr (RC) {
        sv[0] = Z0[0]; sv[1] = Z0[1]; sv[2] = Z0[2]; sv[3] = Z0[3];
        sv[4] = Z1[0]; sv[5] = Z1[1]; sv[6] = Z1[2]; sv[7] = Z1[3];
      }
      float pvs[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) pvs[r] = __builtin_amdgcn_exp2f(fmaf(sv[r], LOG2E, -lse2_q));   // softmax probability (csa_models.py:141)
      if (last_tile) {
#pragma unroll
        for (int r = 0; r < 8; ++r) pvs[r] = r < nv ? pvs[r] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float pv = pvs[r];
        const float md = (!drop || keep[r]) ? keep_scale : 0.f;    // d P_drop / d P
        const float ds = pv * (t1[r] * md - delta_q);              // d softmax (delta = rowsum(dO * O) already has the mask)
        sv[r] = pv * md;                                           // what the dV product needs: the dropped probabilities
        t1[r] = ds;
      }
      if (!RC && !p.sc_tiles) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          csn_bstore4_stream(f32x4{sv[4 * j], sv[4 * j + 1], sv[4 * j + 2], sv[4 * j + 3]}, Sr, s_voff[j]);
          csn_bstore4_stream(f32x4{t1[4 * j], t1[4 * j + 1], t1[4 * j + 2], t1[4 * j + 3]}, dSr, s_voff[j]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      ph[r] = to16<PR::HALF>(t1[r]);
      pl[r] = PR::NT == 3 ? to16<PR::HALF>(t1[r] - from16<PR::HALF>(ph[r])) : ph[r];
    }
    if (BWD && p.sc_tiles) {
      // P and dS leave as tile planes, per query row 16 tiles of [hi: 32 keys | lo: 32 keys] — the k-major operand
      // the dV / dK products stage without conversion work; keys beyond the block end are written as zeros.
      s16x8 qh, ql;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        qh[r] = to16<PR::HALF>(sv[r]);
        ql[r] = PR::NT == 3 ? to16<PR::HALF>(sv[r] - from16<PR::HALF>(qh[r])) : qh[r];
      }
      if constexpr (NPL == 2) {
        // two planes: a row of 16 tiles is exactly the bytes of the fp32 score row — P overwrites the scores of this tile
        // (already consumed by this wave) in place, dS takes the same place in `dscores`
        const unsigned tv = !q_ok ? CSN_OOB
                            : tile_major ? (unsigned)(kt * Tq + qrow) * 128u + (unsigned)(16 * kq)
                                         : (unsigned)(qrow * Tp) * 4u + (unsigned)(kt * 128 + 16 * kq);
        csn_bstore4_stream(__builtin_bit_cast(f32x4, qh), Sr, tv);
        csn_bstore4_stream(__builtin_bit_cast(f32x4, ql), Sr, tv, 64u);
        csn_bstore4_stream(__builtin_bit_cast(f32x4, ph), dSr, tv);
        csn_bstore4_stream(__builtin_bit_cast(f32x4, pl), dSr, tv, 64u);
      } else {
        // one plane: a row is half the bytes, so compact rows cannot overwrite the scores in place (they would run over the
        // rows of other work-groups).  Both go to this block's region of `dscores`: [P: Tq rows | dS: Tq rows] of pitch Tp
        // 16-bit elements; the scores stay untouched
        const unsigned tv = q_ok ? (unsigned)(qrow * Tp) * 2u + (unsigned)(kt * 64 + 16 * kq) : CSN_OOB;
        csn_bstore4_stream(__builtin_bit_cast(f32x4, qh), dSr, tv);
        csn_bstore4_stream(__builtin_bit_cast(f32x4, ph), dSr, tv, (unsigned)(Tq * Tp) * 2u);
      }
    }
  };
  // phase 2: OUT[c][q] += sum_key tileB[c][key] T1[key][q]
  auto phase2 = [&](int st) {
    const lds_s16* tBh = opaque_lds(tileB(st, 0) + b_pos);
    const lds_s16* tBl = tBh + (NPL - 1) * PLANE;
    constexpr int NC = D / 16;
    s16x8 vh[PD], vl[PD];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < PD && c < NC; ++c) {
      vh[c] = *reinterpret_cast<const lds_s16x8*>(tBh + c * 16 * KT);
      vl[c] = *reinterpret_cast<const lds_s16x8*>(tBl + c * 16 * KT);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, NPL * (PD < NC ? PD : NC), 0);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int r = c % PD;
      O[c] = mma16<PR>(vh[r], vl[r], ph, pl, O[c]);
      if (c + PD < NC) {
        vh[r] = *reinterpret_cast<const lds_s16x8*>(tBh + (c + PD) * 16 * KT);
        vl[r] = *reinterpret_cast<const lds_s16x8*>(tBl + (c + PD) * 16 * KT);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, PR::NT, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NPL, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  if constexpr (RC) {
    // three images: V (k-major, tileA), K (k-major, tileC) and K (key-contiguous, tileB).  Both k-major images are read in
    // segment 1, so both are committed in segment 1 of the tile before (see the hazard note below); the K pieces stay in
    // registers (g2) until the key-contiguous image has taken them in segment 2 — or, EARLY, go into the third stage of that
    // image in segment 1 as well (stage (kt + 1) % 3 was last read two tiles ago), which frees both register sets for the
    // request of tile kt + 2 a whole segment earlier.
    fetch_to(Ar, 0, g); commitA_to(tileA(0, 0), tileA(0, NPL - 1), g);
    fetch_to(Br, 0, g2); commitA_to(tileC(0, 0), tileC(0, NPL - 1), g2); commitB_from(0, g2, false);
    if (nkt > 1) { fetch_to(Ar, 1, g); fetch_to(Br, 1, g2); }
  } else if constexpr (EARLY) {
    fetch_to(Ar, 0, g); commitA_to(tileA(0, 0), tileA(0, NPL - 1), g);
    fetch_to(Br, 0, g2); commitB_from(0, g2);
    if (nkt > 1) fetch_to(Ar, 1, g);
  } else {
    PSTAMP(4);
    if constexpr (PRE0) {
      commitA(0); commitB_from(0, gp);
    } else {
      fetch(Ar, 0); commitA(0);
      PSTAMP(5);
      fetch(Br, 0); commitB(0);
    }
    PSTAMP(6);
    if (nkt > 1) fetch(Ar, 1);
  }
  __syncthreads();
  PSTAMP(7);

  // -DCSN_STAMPS: development build that records s_memtime at the phase boundaries of tiles 4..7 (scripts/attn_stamps.py)
#ifdef CSN_STAMPS
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (dbg_on) stamps[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
  unsigned long long stamps[8];
#else
#define STAMP(i)
#endif
  // Staggered schedule.  A tile is two barrier-delimited segments,
  //   seg 1: phase 1 (matrix)              + commit K tile kt+1, request V tile kt+1
  //   seg 2: pointwise (vector) + phase 2  + commit V tile kt+1, request K tile kt+2
  // and waves 4..7 — the SIMD partners of waves 0..3 — run ONE SEGMENT BEHIND (one extra barrier before their loop,
  // one after the loop of the first half): a SIMD then always has one wave in the matrix-only segment beside one that
  // starts with its vector work, instead of two waves fighting over the same pipe.  With the commits placed as above the
  // shift is hazard-free: a K stage is rewritten in segments 2kt / 2kt+1 (early / late half), last read in 2kt-1 and
  // next read in 2kt+2; a V stage is rewritten in 2kt+1 / 2kt+2, last read in 2kt and next read in 2kt+3.
  WGSTAMP(1);
  // Static priority for the late half.  The stamps of the d = 256 forward (profiles/r4u_attention_forward_stamps.txt): waves 4..7
  // — dispatched second, the losers of the SIMD's age-ordered issue arbitration — spend 5.65 k cycles of a tile working and
  // 0.5 k at barriers, waves 0..3 4.4 k and 2.1 k: the older half waits for the younger at both barriers of every tile.
  // Measured with s_setprio 1 on the late half: forward 5.26 / 5.20 ms against 5.25 / 5.17, backward 7.32 / 7.16 against 7.05 /
  // 7.32 — nothing (profiles/r4u_attention_prologue_and_priority.txt).  Off.
#ifndef CSN_LATE_PRIO
#define CSN_LATE_PRIO 0
#endif
  if (late) {
    if (CSN_LATE_PRIO) __builtin_amdgcn_s_setprio(CSN_LATE_PRIO);
    __syncthreads();
  }
  // (a two-tiles-per-trip form of this loop, with the LDS stage a compile-time constant, was built and dropped: at d = 256 it
  //  spilled 28 registers in the forward and 82 in the backward kernel)
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1, nxt = cur ^ 1;
    const int b_cur = (RC && EARLY) ? kt % 3 : cur, b_nxt = (RC && EARLY) ? (kt + 1) % 3 : nxt;     // stage of the key-contiguous image
    const bool more = kt + 1 < nkt;
#ifdef CSN_STAMPS
    const bool dbg_on = (BWD == (CSN_STAMPS != 0)) && DT == 8 && blockIdx.x < 2048 && kt >= 4 && kt < 8;   // -DCSN_STAMPS=0: forward, =1: backward
#endif
    STAMP(0);
    // (requesting the NEXT tile's saved scores right after this tile's pointwise — a phase earlier — was measured in the
    //  backward: 16 more spilled registers inside the loop, 6.8 -> 9.4 ms; the request stays at the top of its own trip)
    load_sv(kt);
    phase1(cur);
    STAMP(1);
    if constexpr (RC) {
      if (more) {
        commitA_to(tileA(nxt, 0), tileA(nxt, NPL - 1), g); commitA_to(tileC(nxt, 0), tileC(nxt, NPL - 1), g2);
        if constexpr (EARLY) {
          commitB_from(b_nxt, g2, false);
          if (kt + 2 < nkt) { fetch_to(Ar, kt + 2, g); fetch_to(Br, kt + 2, g2); }
        }
      }
    } else if constexpr (EARLY) {
      if (more) {
        commitA_to(tileA(nxt, 0), tileA(nxt, NPL - 1), g);
        fetch_to(Br, kt + 1, g2);
        if (kt + 2 < nkt) fetch_to(Ar, kt + 2, g);            // a whole tile before its commit
      }
    } else {
      if (more) { commitA(nxt); fetch(Br, kt + 1); }
    }
    STAMP(2);
    __syncthreads();
    STAMP(3);
    pointwise(kt);
    STAMP(4);
    phase2(b_cur);
    STAMP(5);
    if constexpr (RC) {
      if constexpr (!EARLY) {
        if (more) { commitB_from(nxt, g2, false); if (kt + 2 < nkt) { fetch_to(Ar, kt + 2, g); fetch_to(Br, kt + 2, g2); } }
      }
    } else if constexpr (EARLY) {
      if (more) commitB_from(nxt, g2);
    } else {
      if (more) { commitB(nxt); if (kt + 2 < nkt) fetch(Ar, kt + 2); }
    }
    STAMP(6);
    __syncthreads();
    STAMP(7);
#ifdef CSN_STAMPS
    if (dbg_on && lane == 0) {
      for (int i = 0; i < 8; ++i) csn_dbg[((blockIdx.x * 8 + wave) * 4 + (kt - 4)) * 8 + i] = stamps[i];
    }
#endif
  }

  if (!late) __syncthreads();                           // pairs with the last barrier of the late half: tiles are idle now
  else if (CSN_LATE_PRIO) __builtin_amdgcn_s_setprio(0);
  if constexpr (!BWD) write_out();                      // (the barrier that ends the next item's prologue separates its tile commits from this)
  }                                                     // next item of the group (its prologue reuses the tiles as staging)

  WGSTAMP(2);
  if constexpr (BWD) write_out();
  WGSTAMP(3);
}

template <typename PR, int DT>
int launch_dt(const CsnAttnArgs& a, bool bwd, hipStream_t st) {
  const long long units = (long long)a.n_blocks * a.H * a.E;
  dim3 grid((unsigned)(((units + 7) / 8) * 8 * (((a.Tq > 0 ? a.Tq : a.T) + 127) / 128)));
  if (a.kv_planes) {
    if (bwd) {
      if constexpr (!PR::HALF) {
        if (a.q2) {                                                 // score recomputation: where three LDS images per stage fit
          if constexpr (csn_attn_recompute_fits(PR::NPL, DT))
            hipLaunchKernelGGL((csn_attn_bf16x3_kernel<PR, DT, true, true, true>), grid, dim3(512), 0, st, a);
          else return -1;
        } else hipLaunchKernelGGL((csn_attn_bf16x3_kernel<PR, DT, true, true>), grid, dim3(512), 0, st, a);
      } else return -1;                                             // fp16: forward only (gradients underflow fp16)
    } else hipLaunchKernelGGL((csn_attn_bf16x3_kernel<PR, DT, false, true>), grid, dim3(512), 0, st, a);
  } else {
    if constexpr (PR::NT == 3) {
      if (bwd) hipLaunchKernelGGL((csn_attn_bf16x3_kernel<PR, DT, true, false>), grid, dim3(512), 0, st, a);
      else hipLaunchKernelGGL((csn_attn_bf16x3_kernel<PR, DT, false, false>), grid, dim3(512), 0, st, a);
    } else return -1;                                               // single-product modes: tile-plane K / V only
  }
  return (int)hipGetLastError();
}

template <typename PR>
int launch_any(const CsnAttnArgs& a, int d, bool bwd, hipStream_t st) {
  if (a.E <= 0 || a.n_blocks <= 0) return 0;
  if ((a.ld & 3) || (a.Tp & 3) || (a.ld_kv & 3)) return -2;
  if ((a.T & 3) && a.kv_planes) return -2;                          // ragged key counts: fp32 K/V maps only
  if ((a.q_shape_stride & 3) || (a.kv_shape_stride & 3)) return -4;
  if (a.sc_tiles && a.Tp < (a.T + 31) / 32 * 32) return -2;
  if (a.sc_layout && (PR::NPL != 2 || !a.kv_planes || a.Tq > 0 || a.tq_arr || a.t_arr || a.Tp < (a.T + 31) / 32 * 32 || (bwd && !a.sc_tiles)))
    return -1;                                                      // tile-major scores: block mode, two planes, tile-plane K / V
  if (a.q2 && (!bwd || !a.kv_planes || (a.q2_shape_stride & 3))) return -1;
  if ((a.r_fmt || a.ctx_fmt || a.q2_fmt || a.out_fmt) && (PR::NPL != 1 || !a.kv_planes)) return -1;   // 16-bit maps: single-product modes
  if (a.out_fmt && a.accumulate) return -1;
  if (a.kv_planes && (a.T > 512 || (a.kv_ld & 7) || (a.kv_shape_stride & 7))) return -2;    // 16 tiles of 32 keys per block
  switch (d) {
    case 32: return launch_dt<PR, 1>(a, bwd, st);
    case 64: return launch_dt<PR, 2>(a, bwd, st);
    case 96: return launch_dt<PR, 3>(a, bwd, st);
    case 128: return launch_dt<PR, 4>(a, bwd, st);
    case 256: return launch_dt<PR, 8>(a, bwd, st);
    default: return -5;
  }
}

int launch_mode(const CsnAttnArgs& a, int d, int mode, bool bwd, hipStream_t st) {
  switch (mode) {
    case 1: return launch_any<Bf16x3>(a, d, bwd, st);
    case 2: return launch_any<Bf16>(a, d, bwd, st);
    case 3: return launch_any<F16>(a, d, bwd, st);
    default: return -1;
  }
}

}  // namespace

int csn_launch_attn_fwd_bf16x3(const CsnAttnArgs& a, int d, int mode, hipStream_t st) {
  return launch_mode(a, d, mode, false, st);
}
int csn_launch_attn_bwd_bf16x3(const CsnAttnArgs& a, int d, int mode, hipStream_t st) { return launch_mode(a, d, mode, true, st); }
