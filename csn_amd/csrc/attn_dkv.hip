// Key-stationary attention backward for the score-recomputing ("flash") data flow: dK and dV of one key/value slot without any
// score-sized tensor (autograd of ScaledDotProductAttention.forward, MID-FC/csa_models.py:138-144, w.r.t. k and v).
//
// attn_bf16x3.hip keeps 16 QUERIES per wave in registers and streams key tiles; this kernel is its mirror image.  A work-group
// owns 128 keys of one (key/value slot, head, block): wave w keeps K^T and V^T of its 16 keys as register operands and the
// dK^T / dV^T accumulators [d][16 keys] of those keys, and the work-group streams 32-query tiles of Qs^T and dO^T through LDS,
// for every evaluation that reads this slot, one after the other (the group loop of the grouped dK / dV product it replaces).
// (One plane at d = 96 — BASELINE configs[4]: a wave owns TWO such 16-key groups, the work-group 256 keys; template parameter G.)
// Per tile and wave, with the key on the matrix instruction's lane:
//   phase 1   S [q][key] = sum_d Qs^T[d][q] K^T[d][key]        dP[q][key] = sum_d dO^T[d][q] V^T[d][key]
//   pointwise P = exp2(S log2e - lse2[q]),  mask regenerated from (seed, position),  dS = P (dP mask/(1-p) - delta[q])
//   phase 2   dV^T[c][key] += sum_q dO^T[c][q] P_drop[q][key]  dK^T[c][key] += sum_q Qs^T[c][q] dS[q][key]
// A lane ends phase 1 with 8 CONSECUTIVE queries of its key (the row permutation of attn_bf16x3.hip), which is the B fragment
// of phase 2 as it stands: P and dS never leave the registers.  lse comes from the forward, delta = rowsum(dO * O) from the dQ
// call that runs before this one (csn_block_attn_bwd_dq_recompute_f32 with probs_tiles = 0).
// LDS: per stage four images — Qs and dO tiles each in the k-major form (phase 1, transposing reads) and the query-contiguous
// form (phase 2, 16-byte reads) — of bf16 hi / lo planes (one plane in the single-product mode), two stages; the 32 lse2 /
// delta values of a tile beside them.  fp32 sources are split while they are staged (once per tile and work-group).
// Schedule: the two barrier segments per tile and the one-segment stagger between waves 0..3 and 4..7 of attn_bf16x3.hip.
#include "csn_common.h"
#include "csn_kernels.h"

#ifdef CSN_DKV_STAMPS
// -DCSN_DKV_STAMPS: development build that records s_memtime at the phase boundaries of tiles 4..7 of 1024 work-groups of the
// d = 96 one-plane instances (scripts/dkv_stamps.py)
__device__ unsigned long long csn_dkv_dbg[1024 * 8 * 4 * 8];
extern "C" __attribute__((visibility("default"))) int csn_dkv_debug_read(void* dst, long long bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(csn_dkv_dbg), bytes); }
#define DSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (dbg_on) dstamps[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DSTAMP(i)
#endif

namespace {

constexpr int QT = 32;               // queries per streamed tile
// keys per work-group = 16 per wave: 128 (8 waves), or 64 (4 waves) where three such work-groups fit a CU (csn_dkv_waves)
constexpr float LOG2E = 1.4426950408889634f;

using namespace csn_mode;
typedef f32x4m f32x4v;
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
typedef short __attribute__((address_space(3))) lds_s16;
typedef s16x8 __attribute__((address_space(3))) lds_s16x8;

CSN_DEVINL const lds_s16* opaque_lds(const short* p) {
  const lds_s16* q = (const lds_s16*)p;
  asm volatile("" : "+v"(q));
  return q;
}

// A copy the compiler cannot see through.  The pieces of a 16-bit map are committed as they are; as a plain copy the compiler
// merges the piece with the load's destination register and carries it around the loop — and a loop-carried copy of a
// register with a load in flight makes it wait for ALL outstanding loads before the barrier (the lesson of the row constants
// below).  Two moves here, after the data has arrived, keep the request of the next tile independent.
CSN_DEVINL u32x2 moved(u32x2 v) {
  u32x2 r;
  asm volatile("v_mov_b32 %0, %1" : "=v"(r[0]) : "v"(v[0]));
  asm volatile("v_mov_b32 %0, %1" : "=v"(r[1]) : "v"(v[1]));
  return r;
}

template <typename PR>
CSN_DEVINL f32x4v mma16(s16x8 ah, s16x8 al, s16x8 bh, s16x8 bl, f32x4v c) {
  if constexpr (PR::NT == 3) {
    c = mfma16<PR::HALF>(al, bh, c);
    c = mfma16<PR::HALF>(ah, bl, c);
  }
  return mfma16<PR::HALF>(ah, bh, c);
}

// QF (one-plane mode): 0 = Qs and dO are fp32 maps; 1 / 2 = 16-bit activation maps, Qs in bf16 / fp16 (a math-mode-3 forward:
// converted to bf16 at the commit) and dO in bf16 — a compile-time property: a format branch inside the tile loop costs the
// kernel its schedule (measured: 3.7 -> 5.2 ms at config-5 geometry)
// Occupancy.  At d = 96 one plane the kernel needs ~175 registers: two waves per SIMD, ONE work-group of 8 waves per CU.  Both
// ways to more resident waves were built and measured at config-5 geometry (scripts/dev/ab_attn.sh, ab_step.sh) and lose:
// the register bound of four waves per SIMD spills 96 registers (4.2 -> 15.5 ms); work-groups of four waves / 64 keys
// (-DCSN_DKV_NW=4: three per CU under a bound of 168 registers) stage every tile for half as many keys and still spill 32
// registers (3.7 -> 6.0 ms).  The default stays 8 waves, two per SIMD.  NARROW (one plane, d <= 64): there the kernel does fit
// 128 registers without spills once the fragment rings are one deep and the phase-2 reads follow the pointwise segment, so
// these instances run under the four-wave bound: two work-groups per CU.
#ifndef CSN_DKV_NW
#define CSN_DKV_NW 8
#endif
#ifndef CSN_DKV_NARROW
#define CSN_DKV_NARROW 1
#endif
constexpr int csn_dkv_waves(int npl, int dt) { return (npl == 1 && dt <= 3) ? CSN_DKV_NW : 8; }
#ifndef CSN_DKV_NARROW_MAXDT
#define CSN_DKV_NARROW_MAXDT 2
#endif
constexpr bool csn_dkv_narrow(int npl, int dt) { return CSN_DKV_NARROW && npl == 1 && dt <= CSN_DKV_NARROW_MAXDT; }
// DR: dropout live (a compile-time property since round 6: as a run-time flag every element of the pointwise segment carried its
// own wave-uniform branch around the keep decision — sixteen branch instructions per tile in a loop that is bound by the
// number of instructions a wave issues)
// G: 16-key groups per wave (round 6).  With one group the loop is bound by the instructions a wave issues per tile — requests,
// commits, barriers, scalar bookkeeping, fragment reads — of which only the pointwise segment and the matrix instructions grow
// with the keys.  A wave that owns TWO groups (32 keys; the work-group 256) reuses every fragment it reads for both, and the
// work-group stages each tile for twice the keys: per key half the LDS reads, commits, requests, barriers and scalar work.
// Costs 88 registers (operands, accumulators, S / dP of the second group): the one-plane d = 96 instance has them.
template <typename PR, int DT, int QF = 0, int NW = 8, bool DR = true, int G = 1>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 3 : (csn_dkv_narrow(PR::NPL, DT) ? 4 : 2)) void csn_attn_dkv_kernel(CsnAttnDkvArgs p) {
  constexpr bool NARROW = csn_dkv_narrow(PR::NPL, DT);
  static_assert(QF == 0 || PR::NPL == 1, "16-bit activation maps: the one-plane mode");
  static_assert(NW == 8 || NW == 4, "work-groups of 8 or 4 waves");
  constexpr int NT = 64 * NW;                           // threads
  constexpr int KW = 16 * NW * G;                       // keys per work-group
  constexpr int NPL = PR::NPL;
  constexpr int D = 32 * DT;
  constexpr int PLANE = D * QT + 32;                    // hi and lo planes 64 bytes out of phase (store banks, attn_bf16x3.hip)
  constexpr int NP_T = (D * 8 + NT - 1) / NT;           // 16-byte pieces of a [D][32] fp32 tile per thread
  constexpr int NST = 2;                                // image stages
  constexpr int IMG_EL = 4 * NST * NPL * PLANE;         // [image: QA, OA, QB, OB][stage][plane]
  constexpr int STAGE_EL = D * KW * 2;                  // prologue / epilogue: a [D][KW keys] fp32 block
  constexpr int BUF_EL = IMG_EL > STAGE_EL ? IMG_EL : STAGE_EL;
  static_assert(2 * BUF_EL + 3 * 64 * 4 <= 160 * 1024, "LDS budget of one CU");
  __shared__ __attribute__((aligned(16))) short tiles[BUF_EL + 3 * 64 * 2];
  auto image = [&](int img, int st, int pl) -> short* { return tiles + ((img * NST + st) * NPL + pl) * PLANE; };
  float* rowc = reinterpret_cast<float*>(tiles + BUF_EL);          // [stage of 3][lse2: 32 | delta: 32]
  float* xbuf = reinterpret_cast<float*>(tiles);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 15, kq = lane >> 4;
  // XCD-aware order (attn_bf16x3.hip): the KC key chunks of a unit stream the same Qs / dO tiles — same residue mod 8
  const int KC = (p.T + KW - 1) / KW;
  const int Y = p.n_blocks * p.H;
  const int L = blockIdx.x, slot8 = L & 7, jj = L >> 3;
  const int kc = jj % KC, u = (jj / KC) * 8 + slot8;
  if (u >= Y * p.n_groups) return;
  const int grp = u / Y, hd = (u % Y) % p.H, blk = (u % Y) / p.H;
  const int it0 = p.grp_off ? p.grp_off[grp] : grp, it1 = p.grp_off ? p.grp_off[grp + 1] : grp + 1;
  const bool short_blk = p.T_last > 0 && blk == p.n_blocks - 1;
  const int T = short_blk ? p.T_last : p.T;                         // queries = keys of this block
  if (kc * KW >= T) return;
  const int ld = p.ld, Tp = p.Tp;
  const int e_first = p.eval_ids ? p.eval_ids[it0] : it0;
  const long long kslot = p.kv_index ? p.kv_index[e_first] : e_first;
  const int nqt = (T + QT - 1) / QT;
  const int n_steps = (it1 - it0) * nqt;
#ifndef CSN_DKV_LOCKSTEP
#define CSN_DKV_LOCKSTEP 0
#endif
#ifndef CSN_DKV_ABL        // timing-only ablations (WRONG results): 1 no tile loads, 2 no pointwise, 4 no phase 2, 8 no phase 1
#define CSN_DKV_ABL 0
#endif
  // (-DCSN_DKV_LOCKSTEP=1, measured: all waves in step, ONE barrier per tile instead of two and no stagger)
  const bool late = !CSN_DKV_LOCKSTEP && __builtin_amdgcn_readfirstlane(wave) >= NW / 2;
  const int col0 = 16 * G * wave + lq;                              // this lane's (first) key inside the chunk; group g: + 16 g
  const int key0 = kc * KW + col0;                                  // ... inside the block

  // ---- register operands K^T, V^T [d][16 keys] from the tile planes -----------------------------------------------
  // The chunk's 128 keys are 4 tiles of [hi 32 | lo 32] (one plane: [32]) per row: 256 NPL bytes.  The work-group fetches the
  // [D][4 tiles] block with 16-byte loads into LDS (the tile buffers are idle) and every lane picks its 16-bit values.
  constexpr int RPC = (KW / 8) * NPL;                               // 16-byte pieces per row of the block
  constexpr int CH_K = (D * RPC + NT - 1) / NT;
  s16x8 Kh[G][DT], Kl[G][DT], Vh[G][DT], Vl[G][DT];
  {
    const int kld = p.kv_ld;
    const int cc = tid % RPC, crow = tid / RPC;                     // piece column, first row (rows + NT / RPC per pass)
    constexpr int RPS = NT / RPC;
    short* sbuf = tiles;                                            // [row][piece ^ swizzle][8 shorts], row pitch KW NPL shorts
    auto stage_planes = [&](const short* base) {
      const long long off = kslot * p.kv_shape_stride + (long long)hd * D * kld + (long long)blk * (512 * NPL) + (long long)kc * (KW * NPL);
      const csn_rsrc_t rs = csn_make_rsrc(base + off, ((long long)(D - 1) * kld + KW * NPL) * 2);
      f32x4 ch[CH_K];
#pragma unroll
      for (int t = 0; t < CH_K; ++t) {
        const int row = crow + RPS * t;
        // (only the tiles that hold keys of this block: the projection writes nothing beyond the block's last 32-key tile)
        const bool ok = row < D && (kc * (KW / 32) + cc / (4 * NPL)) < (T + 31) / 32;
        ch[t] = csn_bload4(rs, ok ? (unsigned)(row * kld * 2 + cc * 16) : CSN_OOB);
      }
#pragma unroll
      for (int t = 0; t < CH_K; ++t) {
        const int row = crow + RPS * t;
        if (row < D) *reinterpret_cast<f32x4*>(&sbuf[row * (KW * NPL) + ((cc ^ (2 * ((row >> 3) & 3))) << 3)]) = ch[t];
      }
    };
    // element (row, local key kl): tile kl >> 5, position kl & 31 -> piece (kl >> 5) * 4 NPL + plane * 4 + ((kl & 31) >> 3)
    auto pick = [&](int row, int plane, int col) {
      const int piece = (col >> 5) * (4 * NPL) + plane * 4 + ((col & 31) >> 3);
      return sbuf[row * (KW * NPL) + (((piece ^ (2 * ((row >> 3) & 3))) << 3) | (col & 7))];
    };
    stage_planes(reinterpret_cast<const short*>(p.k));
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
    for (int s = 0; s < DT; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        Kh[g][s][j] = pick(32 * s + 8 * kq + j, 0, col0 + 16 * g);
        Kl[g][s][j] = pick(32 * s + 8 * kq + j, NPL - 1, col0 + 16 * g);
      }
    __syncthreads();
    stage_planes(reinterpret_cast<const short*>(p.v));
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
    for (int s = 0; s < DT; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        Vh[g][s][j] = pick(32 * s + 8 * kq + j, 0, col0 + 16 * g);
        Vl[g][s][j] = pick(32 * s + 8 * kq + j, NPL - 1, col0 + 16 * g);
      }
    __syncthreads();                                                // the staging block becomes the tile images
    if constexpr (NPL == 1 && !PR::HALF) {
      if (p.kv_f16) {                                               // planes of an fp16 forward: this kernel's products are bf16
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
        for (int s = 0; s < DT; ++s)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            Kh[g][s][j] = Kl[g][s][j] = to16<false>(from16<true>(Kh[g][s][j]));
            Vh[g][s][j] = Vl[g][s][j] = to16<false>(from16<true>(Vh[g][s][j]));
          }
      }
    }
  }

  f32x4v dK[G][D / 16], dV[G][D / 16];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
  for (int c = 0; c < D / 16; ++c) { dK[g][c] = f32x4v{0.f, 0.f, 0.f, 0.f}; dV[g][c] = f32x4v{0.f, 0.f, 0.f, 0.f}; }

  constexpr bool drop = DR;
  const unsigned thr16 = csn_drop_threshold16(p.dropout_p);
  const float keep_scale = drop ? 1.f / (1.f - p.dropout_p) : 1.f;
  const int mp = p.T > Tp ? p.T : Tp;                               // mask pitch of the forward (queries per block vs score pitch)
  const unsigned pw_key = (unsigned)((key0 >> 1) * mp);             // pair index of this lane's key: (key / 2) * mp + query (group g: + 8 g mp)
  const bool key_odd = key0 & 1;                                    // (the same in every group: groups are 16 keys apart)

  // ---- streamed tiles: fp32 [d][32 queries] -> bf16 hi / lo -> LDS, in both forms (the fp32 staging of attn_bf16x3.hip) ----
  const int t_c = tid & 7, t_row = tid >> 3;                        // 16-byte piece of the row (queries 4 c ..), first row (+ RPT i)
  const int t_sw = ((t_row >> 3) ^ t_row) & 1, t_swz = (-((t_row >> 2) & 3)) & 3;
  constexpr int RPT = NT / 8;                                       // tile rows per pass of the work-group's threads
  const bool t_last_ok = ((D * 8) % NT == 0) || (tid + NT * (NP_T - 1) < D * 8);
  const int a_dst = t_row * QT + 4 * (t_c ^ t_sw);
  const int b_dst = t_row * QT + 8 * ((t_c >> 1) ^ t_swz) + 4 * (t_c & 1);
  // The pieces of a tile are split into bf16 hi / lo ONCE, when they are committed to the k-major image (segment 1), and wait
  // in that form (as many registers as the fp32 piece in the two-plane mode, half in the one-plane mode) for the
  // query-contiguous image (segment 2).
  f32x4 gQ[NP_T], gO[NP_T];
  u32x2 hQ[NP_T], hO[NP_T];                                         // the same pieces of 16-bit maps (loaded straight into these: no copies of in-flight registers)
  s16x4 cQh[NP_T], cQl[NP_T], cOh[NP_T], cOl[NP_T];
  // wave 0 also carries the tile's 32 lse / 32 delta values (lanes 0..31 / 32..63), as buffer loads like everything else in
  // the loop: nothing may force a wait on memory between a request and the barrier that follows it
  float gca = 0.f, gcb = 0.f;
  const bool wave0 = __builtin_amdgcn_readfirstlane(wave) == 0;
  // fetch stream: (item, query tile) of the next tile to request — it runs two tiles ahead of the products.  Everything that
  // depends on the item (evaluation id, query slot, base pointers) is scalar and reloaded only when the item changes.
  int f_it = it0, f_qt = 0;
  // 16-bit activation maps (one-plane mode): Qs / dO arrive as 16-bit maps — a piece is 8 bytes and is committed as it is
  // (q_fmt == 2: fp16 bits of a math-mode-3 forward, converted to bf16 at the commit)
  constexpr int q_fmt = QF, o_fmt = QF ? 1 : 0;
  constexpr int q_es = q_fmt ? 2 : 4, o_es = o_fmt ? 2 : 4;
  // Everything that depends on the item — the four buffer descriptors (windows: the whole block of this item's maps) — is
  // built when the item changes; a tile adds one scalar offset (round 6: the per-tile descriptors were ~40 scalar instructions
  // of a loop whose pace is the instruction count).  The hardware range check does not see scalar offsets, so queries beyond
  // the block end are switched off in the lane offset; the row constants need no predicate (their lane offset carries the tile).
  csn_rsrc_t Qr_it, Or_it, Lr_it, Dr_it;
  auto fetch_item = [&]() {
    const int e = p.eval_ids ? p.eval_ids[f_it] : f_it;
    const long long qs = p.q_index ? p.q_index[e] : e;
    const long long head_off = (long long)hd * D * ld + (long long)blk * p.T;
    const long long stat = ((long long)e * p.H + hd) * ((long long)p.n_blocks * p.T) + (long long)blk * p.T;
    const long long win = (long long)(D - 1) * ld + T;                                // elements of a [D][T] window of pitch ld
    Qr_it = csn_make_rsrc(reinterpret_cast<const char*>(p.q) + (qs * p.q_shape_stride + head_off) * q_es, win * q_es);
    Or_it = csn_make_rsrc(reinterpret_cast<const char*>(p.dctx) + ((long long)e * p.ctx_eval_stride + head_off) * o_es, win * o_es);
    Lr_it = csn_make_rsrc(p.lse + stat, (long long)T * 4);
    Dr_it = csn_make_rsrc(p.delta + stat, (long long)T * 4);
  };
  fetch_item();
  const unsigned t_off = (unsigned)(t_row * ld + 4 * t_c);                             // (elements) this thread's piece inside a tile
  auto fetch = [&]() {
    const int q_first = f_qt * QT;                                                     // first query of the tile inside the block
    const unsigned off = (q_first + 4 * t_c) < T ? t_off : CSN_OOB;                   // T % 4 == 0: a piece is all in or all out
#pragma unroll
    for (int i = 0; i < NP_T; ++i) {
      const unsigned o = ((CSN_DKV_ABL & 1) || (i == NP_T - 1 && !t_last_ok)) ? CSN_OOB : off;
      if constexpr (q_fmt != 0) hQ[i] = csn_bload2(Qr_it, o == CSN_OOB ? o : o * 2u, (unsigned)(q_first + RPT * i * ld) * 2u);
      else gQ[i] = csn_bload4(Qr_it, o == CSN_OOB ? o : o * 4u, (unsigned)(q_first + RPT * i * ld) * 4u);
      if constexpr (o_fmt != 0) hO[i] = csn_bload2(Or_it, o == CSN_OOB ? o : o * 2u, (unsigned)(q_first + RPT * i * ld) * 2u);
      else gO[i] = csn_bload4(Or_it, o == CSN_OOB ? o : o * 4u, (unsigned)(q_first + RPT * i * ld) * 4u);
    }
    if (wave0) {                                                                       // (queries beyond the block: 0, by the range check)
      gca = csn_bload(Lr_it, lane < 32 ? (unsigned)(q_first + lane) * 4u : CSN_OOB);
      gcb = csn_bload(Dr_it, lane >= 32 ? (unsigned)(q_first + lane - 32) * 4u : CSN_OOB);
    }
    if (++f_qt == nqt) {                                            // next tile: the first of the next item (if any)
      f_qt = 0;
      if (++f_it < it1) fetch_item();
    }
  };
  auto commit_kmajor = [&](int st) {                                // Qs -> image 0, dO -> image 1
#pragma unroll
    for (int i = 0; i < NP_T; ++i)
      if (i < NP_T - 1 || t_last_ok) {
        if constexpr (q_fmt != 0) {
          if constexpr (q_fmt == 2) cQh[i] = f16x4_to_bf16x4(__builtin_bit_cast(s16x4, hQ[i]));
          else cQh[i] = __builtin_bit_cast(s16x4, moved(hQ[i]));
          cQl[i] = cQh[i];
        } else split4<PR>(gQ[i], cQh[i], cQl[i]);
        if constexpr (o_fmt != 0) {
          cOh[i] = __builtin_bit_cast(s16x4, moved(hO[i]));
          cOl[i] = cOh[i];
        } else split4<PR>(gO[i], cOh[i], cOl[i]);
        *reinterpret_cast<s16x4*>(image(0, st, 0) + a_dst + RPT * QT * i) = cQh[i];
        *reinterpret_cast<s16x4*>(image(1, st, 0) + a_dst + RPT * QT * i) = cOh[i];
        if constexpr (NPL == 2) {
          *reinterpret_cast<s16x4*>(image(0, st, 1) + a_dst + RPT * QT * i) = cQl[i];
          *reinterpret_cast<s16x4*>(image(1, st, 1) + a_dst + RPT * QT * i) = cOl[i];
        }
      }
  };
  auto commit_contig = [&](int st) {                                // Qs -> image 2, dO -> image 3
#pragma unroll
    for (int i = 0; i < NP_T; ++i)
      if (i < NP_T - 1 || t_last_ok) {
        *reinterpret_cast<s16x4*>(image(2, st, 0) + b_dst + RPT * QT * i) = cQh[i];
        *reinterpret_cast<s16x4*>(image(3, st, 0) + b_dst + RPT * QT * i) = cOh[i];
        if constexpr (NPL == 2) {
          *reinterpret_cast<s16x4*>(image(2, st, 1) + b_dst + RPT * QT * i) = cQl[i];
          *reinterpret_cast<s16x4*>(image(3, st, 1) + b_dst + RPT * QT * i) = cOl[i];
        }
      }
  };
  // The row constants sit in a ring of THREE stages and are committed in segment 1 with the k-major images, straight from
  // the load's registers (a two-stage buffer would have to wait for segment 2, behind a register copy of values that have
  // just been requested again — and such a copy makes the compiler wait for the whole request before the barrier).
  auto commit_rowc = [&](int st3) { if (wave0) rowc[st3 * 64 + lane] = lane < 32 ? gca * LOG2E : gcb; };

  // fragment read positions (lane constants; attn_bf16x3.hip)
  const int tr_row = 8 * kq + (lq >> 2);
  const int tr_sw = (kq ^ (lq >> 2)) & 1;
  const int a_pos0 = tr_row * QT + 8 * (lq & 3) + 4 * tr_sw, a_pos1 = tr_row * QT + 8 * (lq & 3) + 4 * (tr_sw ^ 1);
  const int b_pos = lq * QT + 8 * (kq ^ ((-((lq >> 2) & 3)) & 3));

  // Both products of a phase run in ONE loop (the S and dP accumulators, the dV and dK accumulators are independent): the LDS
  // fragment reads of the two images run PD1 / PD2 steps ahead of the matrix instructions in one register ring each, and a phase
  // pays the LDS latency once.  At the narrow head widths a phase is only 2 x D/16 matrix instructions, so what a tile costs
  // is the number of such exposed latencies, not the matrix work.
  // Ring depths.  Round 6 measured deeper rings on the one-plane instances (-DCSN_DKV_PD1=4 -DCSN_DKV_PD2=6: ALL of phase 2's
  // fragments requested before the pointwise segment, 203 registers at d = 96): phase 2 655 -> 584 cycles by the stamps, the
  // pointwise segment behind the twelve reads 436 -> 655, the config-5 step +-0 (profiles/r6_dkv_kernel.txt) — the phases are
  // not waiting for LDS, the SIMD's two waves are waiting for each other's vector and matrix issue.  Two deep stays.
#ifndef CSN_DKV_PD1
#define CSN_DKV_PD1 2
#endif
#ifndef CSN_DKV_PD2
#define CSN_DKV_PD2 2
#endif
  constexpr int PD1 = NARROW ? 1 : (NPL == 1 ? CSN_DKV_PD1 : 2);
  constexpr int PD2 = NARROW ? 1 : (NPL == 1 ? CSN_DKV_PD2 : 2);
  // phase 1: S[q][key] = sum_d Qs^T[d][q] K^T[d][key]  and  dP[q][key] = sum_d dO^T[d][q] V^T[d][key]   (images 0 and 1)
  f32x4v S0[G], S1[G], P0[G], P1[G];                                // phase 1's accumulators: S and dP of the tile, per key group
  auto phase1 = [&](int st) {
#pragma unroll
    for (int g = 0; g < G; ++g) { S0[g] = f32x4v{0.f, 0.f, 0.f, 0.f}; S1[g] = S0[g]; P0[g] = S0[g]; P1[g] = S0[g]; }
    const short* __restrict__ qh_ = image(0, st, 0);
    const short* __restrict__ ql_ = image(0, st, NPL - 1);
    const short* __restrict__ oh_ = image(1, st, 0);
    const short* __restrict__ ol_ = image(1, st, NPL - 1);
    constexpr int NH = 2 * DT;
    s16x8 aqh[PD1], aql[PD1], aoh[PD1], aol[PD1];
    auto rd = [&](const short* th, const short* tl, int h, s16x8& fh, s16x8& fl) {
      const int o = 32 * (h >> 1) * QT + ((h & 1) ? a_pos1 : a_pos0);
      fh = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(th + o)),
                 __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(th + o + 4 * QT)));
      if constexpr (NPL == 2)
        fl = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tl + o)),
                   __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tl + o + 4 * QT)));
      else fl = fh;
    };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < PD1 && h < NH; ++h) { rd(qh_, ql_, h, aqh[h], aql[h]); rd(oh_, ol_, h, aoh[h], aol[h]); }
    __builtin_amdgcn_sched_group_barrier(0x100, 4 * NPL * (PD1 < NH ? PD1 : NH), 0);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int r = h % PD1, sidx = h >> 1;
#pragma unroll
      for (int g = 0; g < G; ++g) {                                 // (a fragment serves every key group of the wave)
        if (h & 1) { S1[g] = mma16<PR>(aqh[r], aql[r], Kh[g][sidx], Kl[g][sidx], S1[g]); P1[g] = mma16<PR>(aoh[r], aol[r], Vh[g][sidx], Vl[g][sidx], P1[g]); }
        else { S0[g] = mma16<PR>(aqh[r], aql[r], Kh[g][sidx], Kl[g][sidx], S0[g]); P0[g] = mma16<PR>(aoh[r], aol[r], Vh[g][sidx], Vl[g][sidx], P0[g]); }
      }
      if (h + PD1 < NH) { rd(qh_, ql_, h + PD1, aqh[r], aql[r]); rd(oh_, ol_, h + PD1, aoh[r], aol[r]); }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * PR::NT * G, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4 * NPL, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // phase 2: dV^T[c][key] += sum_q dO^T[c][q] P_drop[q][key]  and  dK^T[c][key] += sum_q Qs^T[c][q] dS[q][key]   (images 3 and 2).
  // Its first fragment reads do not depend on the pointwise segment: they are issued before it (phase2_ahead) and land under it.
  constexpr int NC = D / 16;
  s16x8 voh[PD2], vol[PD2], vqh[PD2], vql[PD2];
  const lds_s16* tOh = nullptr; const lds_s16* tQh = nullptr;
  auto phase2_ahead = [&](int st) {
    tOh = opaque_lds(image(3, st, 0) + b_pos);
    tQh = opaque_lds(image(2, st, 0) + b_pos);
#pragma unroll
    for (int c = 0; c < PD2 && c < NC; ++c) {
      voh[c] = *reinterpret_cast<const lds_s16x8*>(tOh + c * 16 * QT);
      vol[c] = *reinterpret_cast<const lds_s16x8*>(tOh + (NPL - 1) * PLANE + c * 16 * QT);
      vqh[c] = *reinterpret_cast<const lds_s16x8*>(tQh + c * 16 * QT);
      vql[c] = *reinterpret_cast<const lds_s16x8*>(tQh + (NPL - 1) * PLANE + c * 16 * QT);
    }
  };
  s16x8 ph[G], pl[G], dh[G], dl[G];                                 // P_drop and dS of the tile as phase-2 B fragments, per key group
  auto phase2 = [&]() {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int r = c % PD2;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        dV[g][c] = mma16<PR>(voh[r], vol[r], ph[g], pl[g], dV[g][c]);
        dK[g][c] = mma16<PR>(vqh[r], vql[r], dh[g], dl[g], dK[g][c]);
      }
      if (c + PD2 < NC) {
        voh[r] = *reinterpret_cast<const lds_s16x8*>(tOh + (c + PD2) * 16 * QT);
        vol[r] = *reinterpret_cast<const lds_s16x8*>(tOh + (NPL - 1) * PLANE + (c + PD2) * 16 * QT);
        vqh[r] = *reinterpret_cast<const lds_s16x8*>(tQh + (c + PD2) * 16 * QT);
        vql[r] = *reinterpret_cast<const lds_s16x8*>(tQh + (NPL - 1) * PLANE + (c + PD2) * 16 * QT);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * PR::NT * G, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * NPL, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- pointwise: this lane's key against queries qt * 32 + 8 kq .. + 7 of tile c_qt (row constants in ring stage rc_cur) ----
  auto pointwise = [&](int c_qt, unsigned salt, int rc_cur) __attribute__((always_inline)) {
    const f32x4 l0 = *reinterpret_cast<const f32x4*>(&rowc[rc_cur * 64 + 8 * kq]), l1 = *reinterpret_cast<const f32x4*>(&rowc[rc_cur * 64 + 8 * kq + 4]);
    const f32x4 d0 = *reinterpret_cast<const f32x4*>(&rowc[rc_cur * 64 + 32 + 8 * kq]), d1 = *reinterpret_cast<const f32x4*>(&rowc[rc_cur * 64 + 32 + 8 * kq + 4]);
    const float lse2[8] = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    const float dlt[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
    const int q0 = c_qt * QT + 8 * kq;                              // first of this lane's 8 queries (inside the block)
#pragma unroll
    for (int g = 0; g < G; ++g) {                                   // (the row constants serve every key group)
    const float sv[8] = {S0[g][0], S0[g][1], S0[g][2], S0[g][3], S1[g][0], S1[g][1], S1[g][2], S1[g][3]};
    const float dp[8] = {P0[g][0], P0[g][1], P0[g][2], P0[g][3], P1[g][0], P1[g][1], P1[g][2], P1[g][3]};
    // Queries beyond the block end (the last tile) need no masking: their rows of Qs and dO are zeros in the images (switched
    // off in the request) and so are their row constants, hence S = dP = delta = 0, P = exp2(0 - 0) = 1 is finite, dS =
    // P (0 - 0) = 0 adds nothing to dK, and P_drop meets a zero row of dO in dV.  (Round 6: the masks were 24 vector
    // instructions in every tile.)
    float pd[8], ds[8];
    // One mixer round decides the two keys of a pair (low / high 16 bits), and the two keys of a pair sit on NEIGHBOURING lanes
    // here: the even lane hashes queries 0..3, the odd lane 4..7, and a quad swap hands each the other's four — half the
    // quarter-rate multiplies of the pointwise segment for four full-rate moves.
    unsigned hsh[8];
    if (drop && !(CSN_DKV_ABL & 2)) {
      unsigned mine[4], theirs[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) mine[j] = csn_pair_hash(pw_key + (unsigned)(8 * g * mp) + (unsigned)(q0 + (key_odd ? 4 : 0) + j), salt);
#pragma unroll
      for (int j = 0; j < 4; ++j) theirs[j] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine[j], 0xB1, 0xf, 0xf, false);   // lane ^ 1
#pragma unroll
      for (int j = 0; j < 4; ++j) { hsh[j] = key_odd ? theirs[j] : mine[j]; hsh[4 + j] = key_odd ? mine[j] : theirs[j]; }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (CSN_DKV_ABL & 2) { pd[r] = sv[r]; ds[r] = dp[r]; continue; }
      const float pv = __builtin_amdgcn_exp2f(fmaf(sv[r], LOG2E, -lse2[r]));  // softmax probability (csa_models.py:141)
      bool keep = true;
      if (drop) keep = (key_odd ? (hsh[r] >> 16) : (hsh[r] & 0xffffu)) >= thr16;
      const float md = keep ? keep_scale : 0.f;                              // d P_drop / d P
      pd[r] = pv * md;                                                       // what dV contracts: the dropped probabilities
      ds[r] = pv * fmaf(dp[r], md, -dlt[r]);                                 // d softmax (the fused form, written out: no contraction choice left to the compiler)
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      ph[g][r] = to16<PR::HALF>(pd[r]);
      pl[g][r] = PR::NT == 3 ? to16<PR::HALF>(pd[r] - from16<PR::HALF>(ph[g][r])) : ph[g][r];
      dh[g][r] = to16<PR::HALF>(ds[r]);
      dl[g][r] = PR::NT == 3 ? to16<PR::HALF>(ds[r] - from16<PR::HALF>(dh[g][r])) : dh[g][r];
    }
    }
  };

  // images: 0 = Qs k-major, 1 = dO k-major, 2 = Qs query-contiguous, 3 = dO query-contiguous
  fetch();
  commit_kmajor(0); commit_contig(0); commit_rowc(0);
  if (n_steps > 1) fetch();
  __syncthreads();
  if (late) __syncthreads();

  // Two barrier segments per tile; waves 4..7 run one segment behind.  The k-major images are read in segment 1 and rewritten
  // in segment 1 of the tile before; the query-contiguous images are read in segment 2 and rewritten in
  // segment 2 of the tile before (the hazard analysis of attn_bf16x3.hip with its A / B stages).  Tile t + 2 is requested in
  // segment 1 of tile t, right after the pieces of tile t + 1 have been split and committed — a whole tile of cover for the
  // HBM latency, which the short matrix phases of the narrow head widths cannot give.
  int c_it = it0, c_qt = 0;                                         // (item, query tile) of the products
  int rc_cur = 0;
  unsigned salt = 0;
  for (int step = 0; step < n_steps; ++step) {
    const int cur = step & 1, nxt = cur ^ 1;
    const int rc_nxt = rc_cur == 2 ? 0 : rc_cur + 1;                // stage of the row constants of this tile / the next
    const bool more = step + 1 < n_steps;
    if (c_qt == 0) {                                                // a new item: its mask salt (scalar unit)
      const int e = p.eval_ids ? p.eval_ids[c_it] : c_it;
      salt = csn_block_salt((unsigned long long)(((long long)e * p.H + hd) * p.n_blocks + blk), p.seed);
    }
#ifdef CSN_DKV_STAMPS
    unsigned long long dstamps[8];
    const bool dbg_on = DT == 3 && PR::NPL == 1 && blockIdx.x >= 2048 && blockIdx.x < 3072 && step >= 4 && step < 8;
#endif
    DSTAMP(0);
    if (!(CSN_DKV_ABL & 8)) phase1(cur);                                    // S = Qs K^T (the forward's product, roles transposed), dP = dO V^T
    DSTAMP(1);
    if (more) {
      commit_kmajor(nxt);                                           // (splits the pieces: the fp32 registers are free again)
      commit_rowc(rc_nxt);
      if (step + 2 < n_steps) fetch();                              // a whole tile ahead of its first use
    }
    DSTAMP(2);
    if (!CSN_DKV_LOCKSTEP) __syncthreads();
    DSTAMP(3);
    if constexpr (!NARROW) phase2_ahead(cur);
    pointwise(c_qt, salt, rc_cur);
    DSTAMP(4);
    if constexpr (NARROW) phase2_ahead(cur);                         // (register diet: nothing of phase 2 lives across the pointwise segment)
    if (!(CSN_DKV_ABL & 4)) phase2();                                                       // dV^T += dO^T P_drop,  dK^T += Qs^T dS
    else { dV[0][0][0] += from16<PR::HALF>(ph[0][0]) + from16<PR::HALF>(ph[0][7]); dK[0][0][0] += from16<PR::HALF>(dh[0][0]) + from16<PR::HALF>(dh[0][7]); }
    DSTAMP(5);
    if (more) commit_contig(nxt);
    if (++c_qt == nqt) { c_qt = 0; ++c_it; }
    rc_cur = rc_nxt;
    DSTAMP(6);
    __syncthreads();
    DSTAMP(7);
#ifdef CSN_DKV_STAMPS
    if (dbg_on && lane == 0)
      for (int i = 0; i < 8; ++i) csn_dkv_dbg[(((blockIdx.x - 2048) * 8 + wave) * 4 + (step - 4)) * 8 + i] = dstamps[i];
#endif
  }
  if (!late) __syncthreads();                                       // pairs with the last barrier of the late half

  // ---- epilogue: dK^T, dV^T [d][KW keys] leave as 16-byte rows through an LDS transpose ---------------------------------
  constexpr int RPE = NT / (KW / 4);                                // rows per pass of the work-group's threads (16; two groups: 8)
  const int cc = tid & (KW / 4 - 1), crow = tid / (KW / 4);       // 4-key chunk of the row, first row (+ RPE t)
  constexpr int CH_T = D / RPE;
  const long long okslot = p.dk_index ? p.dk_index[e_first] : e_first, ovslot = p.dv_index ? p.dv_index[e_first] : e_first;
  const long long out_off = (long long)hd * D * ld + (long long)blk * p.T + kc * KW;
  const int nk = T - kc * KW;                                       // keys of this chunk that exist
  const long long owin = ((long long)(D - 1) * ld + (nk < KW ? nk : KW)) * 4;
  const unsigned c_off = (4 * cc) < nk ? (unsigned)(crow * ld + 4 * cc) * 4u : CSN_OOB;
  auto store_out = [&](const f32x4v (*OUT)[D / 16], float* base, long long slot) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int col = col0 + 16 * g;
#pragma unroll
      for (int c = 0; c < D / 16; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * c + 4 * kq + r;
          xbuf[row * KW + ((((col >> 2) ^ (4 * ((row >> 2) & 1))) << 2) | (col & 3))] = OUT[g][c][r];
        }
    }
    __syncthreads();
    f32x4 ch[CH_T];
#pragma unroll
    for (int t = 0; t < CH_T; ++t) {
      const int row = crow + RPE * t;
      ch[t] = *reinterpret_cast<const f32x4*>(&xbuf[row * KW + ((cc ^ (4 * ((row >> 2) & 1))) << 2)]);
    }
    if (NPL == 1 && p.out_fmt) {                                    // bf16 gradient maps (16-bit activation maps; written once)
      const csn_rsrc_t r16 = csn_make_rsrc(reinterpret_cast<short*>(base) + slot * p.dkv_slot_stride + out_off, owin / 2);
#pragma unroll
      for (int t = 0; t < CH_T; ++t)
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, to16x4<false>(ch[t])), r16, c_off == CSN_OOB ? CSN_OOB : c_off >> 1,
                                              (unsigned)(RPE * t * ld) * 2u, 0);
      return;
    }
    const csn_rsrc_t rs = csn_make_rsrc(base + slot * p.dkv_slot_stride + out_off, owin);
    if (p.accumulate) {
      f32x4 prev[CH_T];
#pragma unroll
      for (int t = 0; t < CH_T; ++t) prev[t] = csn_bload4(rs, c_off, (unsigned)(RPE * t * ld) * 4u);
#pragma unroll
      for (int t = 0; t < CH_T; ++t) ch[t] += prev[t];
    }
#pragma unroll
    for (int t = 0; t < CH_T; ++t) csn_bstore4(ch[t], rs, c_off, (unsigned)(RPE * t * ld) * 4u);
  };
  store_out(dK, p.dk, okslot);
  __syncthreads();
  store_out(dV, p.dv, ovslot);
}

#ifndef CSN_DKV_G2
#define CSN_DKV_G2 1
#endif
// two 16-key groups per wave where the registers allow: one plane, d = 96 (the instances below it run two work-groups per CU
// under the four-wave bound instead, the ones above it and the two-plane ones have no room)
constexpr int csn_dkv_groups(int npl, int dt, int nw) { return (CSN_DKV_G2 && npl == 1 && dt == 3 && nw == 8) ? 2 : 1; }

template <typename PR, int DT, bool DR>
int launch_dt(const CsnAttnDkvArgs& a, hipStream_t st) {
  constexpr int NW = csn_dkv_waves(PR::NPL, DT);
  constexpr int G = csn_dkv_groups(PR::NPL, DT, NW);
  const long long units = (long long)a.n_blocks * a.H * a.n_groups;
  const int KC = (a.T + 16 * NW * G - 1) / (16 * NW * G);
  dim3 grid((unsigned)(((units + 7) / 8) * 8 * KC));
  if (a.q_fmt || a.dctx_fmt) {
    if constexpr (PR::NPL == 1) {
      if (a.dctx_fmt != 1) return -1;
      if (a.q_fmt == 1) hipLaunchKernelGGL((csn_attn_dkv_kernel<PR, DT, 1, NW, DR, G>), grid, dim3(64 * NW), 0, st, a);
      else if (a.q_fmt == 2) hipLaunchKernelGGL((csn_attn_dkv_kernel<PR, DT, 2, NW, DR, G>), grid, dim3(64 * NW), 0, st, a);
      else return -1;
    } else return -1;
  } else hipLaunchKernelGGL((csn_attn_dkv_kernel<PR, DT, 0, NW, DR, G>), grid, dim3(64 * NW), 0, st, a);
  return (int)hipGetLastError();
}

template <typename PR, int DT>
int launch_dt(const CsnAttnDkvArgs& a, hipStream_t st) {
  return a.dropout_p > 0.f ? launch_dt<PR, DT, true>(a, st) : launch_dt<PR, DT, false>(a, st);
}

template <typename PR>
int launch_any(const CsnAttnDkvArgs& a, int d, hipStream_t st) {
  switch (d) {
    case 32: return launch_dt<PR, 1>(a, st);
    case 64: return launch_dt<PR, 2>(a, st);
    case 96: return launch_dt<PR, 3>(a, st);
    case 128: return launch_dt<PR, 4>(a, st);
    default: return -5;
  }
}

}  // namespace

int csn_launch_attn_dkv_flash(const CsnAttnDkvArgs& a, int d, int mode, hipStream_t st) {
  if (a.n_groups <= 0 || a.n_blocks <= 0) return 0;
  if ((a.ld & 3) || (a.T & 3) || (a.T_last & 3) || a.T > 512 || (a.kv_ld & 7) || (a.kv_shape_stride & 7)) return -2;
  if ((a.q_shape_stride & 3) || (a.ctx_eval_stride & 3) || (a.dkv_slot_stride & 3)) return -4;
  if ((a.q_fmt || a.dctx_fmt) && mode != 2) return -1;              // 16-bit activation maps: the one-plane mode
  if (a.dctx_fmt == 2) return -1;                                   // dO is a backward tensor: bf16
  if (a.out_fmt && (mode != 2 || a.accumulate)) return -1;
  switch (mode) {
    case 1: return launch_any<Bf16x3>(a, d, st);
    case 2: return launch_any<Bf16>(a, d, st);
    default: return -1;
  }
}
