// Batched GEMM with fp32 operands emulated by three bf16 matrix-core products ("bf16x3").
//
//   x = hi + lo,  hi = bf16(x),  lo = bf16(x - hi)          (|x - hi - lo| <= 2^-17 |x|)
//   a * b  ~=  a_hi b_hi + a_hi b_lo + a_lo b_hi             (fp32 accumulate in the MFMA)
//
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the exact fp32 matrix instruction, so three of them per
// product are 5.3x faster than v_mfma_f32_32x32x2_f32 at ~1e-5 relative error per product — inside the
// 1e-4 contract of the CSA path (measured end to end: logits 2e-6, weight gradients <= 1e-5 relative).
// This is the "fast" math mode (csn_set_math_mode(1)); the exact fp32 kernels of gemm_f32.hip stay the
// reference mode.  Same interface and operand layouts as gemm_f32.hip:
//   A : MK (k contiguous)             -> fragments by one 16-byte LDS read per lane
//   B : NK (k contiguous per column)  -> the same
//       KN (k-major, lanes along n)   -> fragments by ds_read_b64_tr_b16, the CDNA4 transposing LDS read
// Operands arrive as fp32 and are split into their bf16 hi / lo planes while they are staged into LDS (two bf16
// planes take exactly the bytes of the fp32 tile).  The split of slab k+1 (VALU) is issued between the matrix
// instructions of slab k (two LDS stages, one barrier per slab), so it overlaps with the matrix pipe.  C can be written
// as fp32 or as bf16 planes (CsnOperand::planes).  (Reading pre-split planes was tried and measured slower on this
// path: blocks of 500 points are only 8-byte aligned in bf16, which halves the width of every staging load.)
//
// MFMA fragment maps (32x32x16 bf16): A lane l holds A[l & 31][8 (l >> 5) + j], B lane l holds
// B[8 (l >> 5) + j][l & 31], j = 0..7; C/D as for the fp32 shape.
// SINGLE-PRODUCT MODES (math modes 2 and 3): the same kernels with the lo planes dropped — every operand is rounded once to
// bf16 (mode 2) or fp16 (mode 3) and a product is ONE matrix instruction (~2^-9 / 2^-11 relative error per operand: outside
// the 1e-4 contract, reported with its measured error).  Tile planes then hold one plane: per row and block 16 tiles of
// 32 elements (block pitch 512 instead of 1024).  Template parameter PR = csn_mode::{Bf16x3, Bf16, F16} (csn_common.h).
#include "csn_common.h"
#include "csn_kernels.h"
#include "csn_window.h"

// -DCSN_STAMPS: development build that records s_memtime after the prologue, the main loop and the epilogue of every
// work-group (scripts/gemm_stamps.py)
#ifdef CSN_STAMPS
__device__ unsigned long long csn_gdbg[65536 * 8];
extern "C" __attribute__((visibility("default"))) int csn_gemm_debug_read(void* dst, long long bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(csn_gdbg), bytes); }
#define GSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); gst[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
// 256 x 256 kernel: entry, tile loop start, tile loop end, [LN: after the residual pass, after the variance pass], exit
#define BSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && blockIdx.x < 8192) csn_gdbg[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GSTAMP(i)
#define BSTAMP(i)
#endif

namespace {

using namespace csn_mode;

constexpr int BK = 32;                 // two MFMA k-steps of 16
constexpr int PK = BK + 8;             // pitch of k-contiguous planes (16-bit elements): 80-byte rows, conflict-free b128 reads

// acc += a * b on the 32x32x16 matrix instruction: three products of the hi / lo terms (small terms first), or one
template <typename PR>
CSN_DEVINL f32x16 mma32(s16x8 ah, s16x8 al, s16x8 bh, s16x8 bl, f32x16 c) {
  if constexpr (PR::NT == 3) {
    c = mfma32<PR::HALF>(al, bh, c);
    c = mfma32<PR::HALF>(ah, bl, c);
  }
  return mfma32<PR::HALF>(ah, bh, c);
}

// BT: the k-major B operand arrives as bf16 "tile planes" (see attn_bf16x3.hip): per k row, consecutive 32-column tiles of
// [hi 32 | lo 32] (B.ld = row pitch in bf16 elements, B strides in bf16 elements; padding columns inside a tile are zero).
// Its staging is then a plain copy — 16-byte loads, 16-byte LDS stores, no conversion work.  This is how the attention
// backward hands the probabilities P and the score gradients dS to the dV / dK products.
// AF / BF: input formats of A / B in the single-product modes (see the 256 x 256 kernel below)
template <typename PR, int BM, int BN, bool B_NK, bool BT = false, int AF = 0, int BF = 0>
__global__ __launch_bounds__(256, 2) void csn_gemm_bf16x3_kernel(CsnGemmArgs p) {
  constexpr int NPL = PR::NPL;                          // planes per operand: hi (+ lo)
  static_assert(!BT || (!B_NK && BN == 128), "tile-plane B: k-major, 128-column tiles");
  static_assert((AF == 0 && BF == 0) || PR::NPL == 1, "16-bit operand maps: single-product modes only");
  static_assert(!(BT && BF) && !((AF == 2 || BF == 2) && PR::HALF), "tile planes carry their own format; fp16 -> bf16 feeds bf16 products");
  constexpr int AES = AF ? 2 : 4, BES = BF ? 2 : 4;
  constexpr int A16_PASS = BM / 64, B16_PASS = B_NK ? BN / 64 : 2;    // 16-byte pieces per thread and slab of a 16-bit map
  constexpr int TPR16 = BN / 8;                         // k-major 16-bit B: threads per k row (16 k rows per pass)
  static_assert(!BF || B_NK || 256 / TPR16 == 16, "k-major 16-bit B: two passes of 16 k rows");
  constexpr int MT = BM / 64, NT = BN / 64;
  constexpr int A_PASS = BM / 32, B_PASS = BN / 32;
  constexpr int TPR = BN / 4, RPP = 256 / TPR;          // KN staging: threads per k row, k rows per pass
  constexpr int PN = BN + 32;                           // pitch of the k-major B planes: rows 64 B apart mod 256
  constexpr int A_EL = BM * PK, B_EL = B_NK ? BN * PK : BK * PN;
  __shared__ __attribute__((aligned(16))) short As[2][NPL][A_EL];   // [stage][plane][row][k]
  __shared__ __attribute__((aligned(16))) short Bs[2][NPL][B_EL];   // NK: [stage][plane][col][k]   KN: [stage][plane][k][col]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef CSN_STAMPS
  unsigned long long gst[4], gin[4] = {0, 0, 0, 0}, gt0 = 0, gt1 = 0;
#endif
  GSTAMP(0);
  const int l31 = lane & 31, h = lane >> 5;
  const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);

  // XCD-aware work-group order: the tiles of one batch item z share its A and B panels, so they should share one
  // XCD's L2.  Work-groups are dealt round-robin over the 8 XCDs; tile t of item z gets id 8 * (tiles * (z / 8) + t) + z % 8.
  const int tiles_n = (p.N + BN - 1) / BN, tiles = tiles_n * ((p.M + BM - 1) / BM);
  const int L = blockIdx.x, jj = L >> 3;
  const int tile = jj % tiles;
  int z = (jj / tiles) * 8 + (L & 7);
  if (z >= p.batch) return;
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  const int z0 = z % p.n0; z /= p.n0;
  const int z1 = z % p.n1;
  const int z2 = p.eval_ids ? p.eval_ids[z / p.n1] : z / p.n1;
  const int lda = p.A.ld, ldb = p.B.ld, ldc = p.C.ld;
  const int M = p.M, N = (p.n_last > 0 && z0 == p.n0 - 1) ? p.n_last : (p.n_arr ? ((p.n_arr[z2] + 3) & ~3) : p.N);   // ragged batches: this item's own columns (rounded up to 4) / contraction count
  int K = (p.k_last > 0 && z0 == p.n0 - 1) ? p.k_last : (p.k_arr ? p.k_arr[z2] : p.K);
  if (p.k_chunk > 0) { K = min(p.k_chunk, p.K - z0 * p.k_chunk); }
  if (tile_n * BN >= N) return;                                // (a tile beyond a short item's columns; before any barrier)                                // (a tile beyond a short item's columns; before any barrier)
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const bool c_pl = p.C.planes != 0;
  const bool c_tiles = p.C.planes == 2;      // "tile planes" for the attention kernels (attn_bf16x3.hip), block = plane_stride points
  const int c_es = c_pl ? 2 : 4;
  const char* a_base = reinterpret_cast<const char*>(p.A.ptr) +
                       (p.A.s0 * z0 + p.A.s1 * z1 + p.A.s2 * (long long)(p.A.idx2 ? p.A.idx2[z2] : z2) + (long long)m0 * lda) * AES;
  const long long b_el = p.B.s0 * z0 + p.B.s1 * z1 + p.B.s2 * (long long)(p.B.idx2 ? p.B.idx2[z2] : z2);
  const char* b_base = BT ? reinterpret_cast<const char*>(reinterpret_cast<const short*>(p.B.ptr) + b_el)
                          : reinterpret_cast<const char*>(p.B.ptr) + (b_el + (B_NK ? (long long)n0 * ldb : (long long)n0)) * BES;
  char* c_base = reinterpret_cast<char*>(p.C.ptr) + (p.C.s0 * z0 + p.C.s1 * z1 + p.C.s2 * (long long)(p.C.idx2 ? p.C.idx2[z2] : z2) + (long long)m0 * ldc + (c_tiles ? 0 : n0)) * c_es;
  const long long c_win = (long long)BM * ldc * c_es;
  // (16-bit k-contiguous maps: the window ends with the last valid row's K elements, see the 256 x 256 kernel)
  const csn_rsrc_t Ar = csn_make_rsrc(a_base, AF ? csn_kwin_bytes(min(BM, M - m0), lda, K, AES) : (long long)BM * lda * AES);
  const csn_rsrc_t Br = csn_make_rsrc(b_base, BT ? (long long)K * ldb * 2
                                                 : (B_NK ? (BF ? csn_kwin_bytes(min(BN, N - n0), ldb, K, BES) : (long long)BN * ldb * BES)
                                                         : ((long long)(K - 1) * ldb + (N - n0)) * BES));
  const csn_rsrc_t Cr = csn_make_rsrc(c_base, c_win), Crl = csn_make_rsrc(c_base + p.C.plane_stride * 2, c_pl ? c_win : 0);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int pr = tid >> 3, pc = (tid & 7) * 4;
  const int kr = tid / TPR, kc = (tid % TPR) * 4;
  // tile-plane B: a k row of the tile is B_PASS * NPL / 2 ... contiguous 16-byte units w = (tid & 7) + 8 i of NPL * 64-byte
  // tiles; unit w sits in tile w / (4 NPL), plane (w % (4 NPL)) / 4, keys 8 (w % 4) .. + 7 of the tile
  constexpr int BT_PASS = B_PASS * NPL / 2;             // 16-byte units per thread and slab
  // 16-bit maps: k-contiguous row pr2 + 64 i, 16-byte unit pu of its 4; k-major B: k row kr2 + 16 i, columns 8 ku .. + 7
  const int pr2 = tid >> 2, pu = tid & 3;
  const int kr2 = tid / TPR16, ku = tid % TPR16;
  unsigned a_off[A_PASS], b_off[B_PASS];
  int bt_dst[B_PASS];
#pragma unroll
  for (int i = 0; i < A_PASS; ++i) {
    if (AF) a_off[i] = (i < A16_PASS && (m0 + pr2 + 64 * i) < M) ? (unsigned)((pr2 + 64 * i) * lda + 8 * pu) * 2u : CSN_OOB;
    else a_off[i] = (m0 + pr + 32 * i) < M ? (unsigned)((pr + 32 * i) * lda + pc) * 4u : CSN_OOB;
  }
#pragma unroll
  for (int i = 0; i < B_PASS; ++i) {
    if (BF) {
      if (B_NK) b_off[i] = (i < B16_PASS && (n0 + pr2 + 64 * i) < N) ? (unsigned)((pr2 + 64 * i) * ldb + 8 * pu) * 2u : CSN_OOB;
      else b_off[i] = (i < B16_PASS && (n0 + 8 * ku) < N) ? (unsigned)((kr2 + 16 * i) * ldb + 8 * ku) * 2u : CSN_OOB;
      bt_dst[i] = 0;
    } else if (BT) {
      const int w = (tid & 7) + 8 * i, tile = w / (4 * NPL), within = w % (4 * NPL), t_u = within & 3;
      b_off[i] = (i < BT_PASS && (n0 + 32 * tile + 8 * t_u) < N)
                     ? (unsigned)(pr * ldb) * 2u + (unsigned)(n0 >> 5) * (unsigned)(64 * NPL) + (unsigned)w * 16u : CSN_OOB;
      bt_dst[i] = (within >> 2) * B_EL + pr * PN + 32 * tile + 8 * t_u;
    } else if (B_NK) b_off[i] = (n0 + pr + 32 * i) < N ? (unsigned)((pr + 32 * i) * ldb + pc) * 4u : CSN_OOB;
    else b_off[i] = (n0 + kc) < N ? (unsigned)((kr + RPP * i) * ldb + kc) * 4u : CSN_OOB;
  }

  f32x4 ra[A_PASS], rb[B_PASS];
  auto load_slab = [&](int k0) {
    const unsigned kp = (k0 + pc) < K ? 0u : CSN_OOB;
    const unsigned kp16 = csn_unit_starts_inside(k0 + 8 * pu, K) ? 0u : CSN_OOB;
    if (AF) {
#pragma unroll
      for (int i = 0; i < A16_PASS; ++i) ra[i] = csn_bload4(Ar, a_off[i] | kp16, (unsigned)k0 * 2u);
    } else {
#pragma unroll
      for (int i = 0; i < A_PASS; ++i) ra[i] = csn_bload4(Ar, a_off[i] | kp, (unsigned)k0 * 4u);
    }
    if (BF) {
#pragma unroll
      for (int i = 0; i < B16_PASS; ++i) {
        if (B_NK) rb[i] = csn_bload4(Br, b_off[i] | kp16, (unsigned)k0 * 2u);
        else rb[i] = csn_bload4(Br, b_off[i] | ((k0 + kr2 + 16 * i) < K ? 0u : CSN_OOB), (unsigned)k0 * (unsigned)ldb * 2u);
      }
    } else if (BT) {
      const unsigned kb = (k0 + pr) < K ? 0u : CSN_OOB;               // k row pr of the slab
#pragma unroll
      for (int i = 0; i < BT_PASS; ++i) rb[i] = csn_bload4(Br, b_off[i] | kb, (unsigned)k0 * (unsigned)ldb * 2u);
    } else if (B_NK) {
#pragma unroll
      for (int i = 0; i < B_PASS; ++i) rb[i] = csn_bload4(Br, b_off[i] | kp, (unsigned)k0 * 4u);
    } else {
#pragma unroll
      for (int i = 0; i < B_PASS; ++i) {
        const unsigned kq = (k0 + kr + RPP * i) < K ? 0u : CSN_OOB;
        rb[i] = csn_bload4(Br, b_off[i] | kq, (unsigned)k0 * (unsigned)ldb * 4u);
      }
    }
  };
  auto unit16 = [&](f32x4 v, int fmt, int k0) {            // (see the 256 x 256 kernel)
    if (fmt == CSN_FMT_F16_TO_BF16) v = f16x8_to_bf16x8(v);
    if (csn_unit_upper_half_beyond(k0 + 8 * pu, K)) { v[2] = 0.f; v[3] = 0.f; }
    return v;
  };
  auto store_slab = [&](int st, int k0) {
    s16x4 hi, lo;
    if (AF) {
#pragma unroll
      for (int i = 0; i < A16_PASS; ++i) *reinterpret_cast<f32x4*>(&As[st][0][(pr2 + 64 * i) * PK + 8 * pu]) = unit16(ra[i], AF, k0);
    } else {
#pragma unroll
      for (int i = 0; i < A_PASS; ++i) {
        split4<PR>(ra[i], hi, lo);
        *reinterpret_cast<s16x4*>(&As[st][0][(pr + 32 * i) * PK + pc]) = hi;
        if constexpr (NPL == 2) *reinterpret_cast<s16x4*>(&As[st][NPL - 1][(pr + 32 * i) * PK + pc]) = lo;
      }
    }
    if (BF) {
#pragma unroll
      for (int i = 0; i < B16_PASS; ++i) {
        if (B_NK) *reinterpret_cast<f32x4*>(&Bs[st][0][(pr2 + 64 * i) * PK + 8 * pu]) = unit16(rb[i], BF, k0);
        else *reinterpret_cast<f32x4*>(&Bs[st][0][(kr2 + 16 * i) * PN + 8 * ku]) = BF == CSN_FMT_F16_TO_BF16 ? f16x8_to_bf16x8(rb[i]) : rb[i];
      }
      return;
    }
    if (BT) {
#pragma unroll
      for (int i = 0; i < BT_PASS; ++i) *reinterpret_cast<f32x4*>(&Bs[st][0][bt_dst[i]]) = rb[i];
      return;
    }
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
      split4<PR>(rb[i], hi, lo);
      const int dst = B_NK ? (pr + 32 * i) * PK + pc : (kr + RPP * i) * PN + kc;
      *reinterpret_cast<s16x4*>(&Bs[st][0][dst]) = hi;
      if constexpr (NPL == 2) *reinterpret_cast<s16x4*>(&Bs[st][NPL - 1][dst]) = lo;
    }
  };

  // transposing read position of this lane for the k-major B planes: group g = lane >> 4 covers columns
  // 16 (g & 1) .. +15 and k rows 8 (g >> 1) .. +7; inside the group lane 4 q + p addresses row q, columns 4 p .. 4 p + 3
  const int grp = lane >> 4, gq = (lane >> 2) & 3, gp = lane & 3;
  const int tr_base = (8 * (grp >> 1) + gq) * PN + 16 * (grp & 1) + 4 * gp;

  const int nk = (K + BK - 1) / BK;
  if (nk > 0) { load_slab(0); store_slab(0, 0); }
  if (nk > 1) load_slab(BK);
  __syncthreads();
  GSTAMP(1);
#ifdef CSN_STAMPS
#define GIN(i, a, b) do { __builtin_amdgcn_sched_barrier(0); a = __builtin_amdgcn_s_memtime(); gin[i] += a - b; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GIN(i, a, b)
#endif
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
#ifdef CSN_STAMPS
    __builtin_amdgcn_sched_barrier(0); gt0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      s16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int o = (wm0 + 32 * i + l31) * PK + 16 * s + 8 * h;
        ah[i] = *reinterpret_cast<const s16x8*>(&As[cur][0][o]);
        al[i] = *reinterpret_cast<const s16x8*>(&As[cur][NPL - 1][o]);          // (one plane: al = ah, unused)
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (B_NK) {
          const int o = (wn0 + 32 * j + l31) * PK + 16 * s + 8 * h;
          bh[j] = *reinterpret_cast<const s16x8*>(&Bs[cur][0][o]);
          bl[j] = *reinterpret_cast<const s16x8*>(&Bs[cur][NPL - 1][o]);
        } else {
          const int o = tr_base + (16 * s) * PN + wn0 + 32 * j;
          typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
          bh[j] = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(&Bs[cur][0][o])),
                        __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(&Bs[cur][0][o + 4 * PN])));
          if constexpr (NPL == 2)
            bl[j] = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(&Bs[cur][1][o])),
                          __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(&Bs[cur][1][o + 4 * PN])));
          else bl[j] = bh[j];
        }
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mma32<PR>(ah[i], al[i], bh[j], bl[j], acc[i][j]);
    }
    // slab kt + 1 (loaded one iteration ago) is split and written into the other stage while the matrix pipe drains;
    // slab kt + 2 starts its trip from HBM
    GIN(0, gt1, gt0);
    if (kt + 1 < nk) {
      store_slab(cur ^ 1, (kt + 1) * BK);
      GIN(1, gt0, gt1);
      if (kt + 2 < nk) load_slab((kt + 2) * BK);
    }
    GIN(3, gt1, gt0);
    __syncthreads();
    GIN(2, gt0, gt1);
  }

  GSTAMP(2);
  const float alpha = p.alpha;
  // output positions: the lane's part (its column, its half's 4-row shift; columns beyond N switched off) in cv[j], the row of
  // accumulator (i, r) as a scalar offset c_so(i, r), rows beyond M cut off by the window Cw — no per-element offset registers
  // (64 of them used to live here, which alone kept the 128 x 128 instances at two waves per SIMD)
  unsigned cv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int nl = wn0 + 32 * j + l31;
    cv[j] = (n0 + nl) < N ? (unsigned)(4 * h * ldc + nl) * (unsigned)c_es : CSN_OOB;
  }
  auto c_so = [&](int i, int r) { return (unsigned)((wm0 + 32 * i + csn_acc_row(r, 0)) * ldc) * (unsigned)c_es; };
  const long long c_rows = min(BM, M - m0);
  const csn_rsrc_t Cw = csn_make_rsrc(c_base, c_rows * ldc * c_es), Cwl = csn_make_rsrc(c_base + p.C.plane_stride * 2, c_pl ? c_rows * ldc * c_es : 0);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = wm0 + 32 * i + csn_acc_row(r, h);
        float v = acc[i][j][r] * alpha;
        if ((m0 + ml) < p.div_rows) v = v / p.div_val;
        acc[i][j][r] = v;
      }
  if (c_tiles) {
    // column n of the map -> block n / Tb, key k = n % Tb -> tile k / 32, position k % 32; a row holds, per block, 16 tiles
    // of [hi: 32 | lo: 32] bf16 (block pitch 1024; the padding keys of a block's last tile are written as zeros)
    const int Tb = (int)p.C.plane_stride;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wn0 + 32 * j + l31;
      const int blk = n / Tb, kib = n - blk * Tb;
      const unsigned col = (unsigned)(blk * (512 * NPL) + (kib >> 5) * (32 * NPL) + (kib & 31));
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ml = wm0 + 32 * i + csn_acc_row(r, h);
          const unsigned off = ((m0 + ml) < M && n < N) ? ((unsigned)(ml * ldc) + col) * 2u : CSN_OOB;
          const short hi = to16<PR::HALF>(acc[i][j][r]);
          csn_bstore16(hi, Cr, off);
          if constexpr (NPL == 2) csn_bstore16(to16<PR::HALF>(acc[i][j][r] - from16<PR::HALF>(hi)), Cr, off, 64u);
          if (off != CSN_OOB && (kib + 1 == Tb || n + 1 == N)) {       // last key of its block: zero padding up to the tile's end
            for (int g = 1; g <= ((31 - (kib & 31)) & 31); ++g) {
              csn_bstore16((short)0, Cr, off + 2u * (unsigned)g);
              if constexpr (NPL == 2) csn_bstore16((short)0, Cr, off + 2u * (unsigned)g, 64u);
            }
          }
        }
    }
    return;
  }
  if (c_pl) {
    // split once here, so that every consumer of C stages plain bf16 planes (no accumulation into planes)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const short hi = to16<PR::HALF>(acc[i][j][r]);
          csn_bstore16(hi, Cw, cv[j], c_so(i, r));
          if constexpr (NPL == 2) csn_bstore16(to16<PR::HALF>(acc[i][j][r] - from16<PR::HALF>(hi)), Cwl, cv[j], c_so(i, r));
        }
    return;
  }
  if (p.accumulate) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f32x16 prev;
#pragma unroll
        for (int r = 0; r < 16; ++r) prev[r] = csn_bload(Cw, cv[j], c_so(i, r));
        acc[i][j] += prev;
      }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) csn_bstore(acc[i][j][r], Cw, cv[j], c_so(i, r));
#ifdef CSN_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  GSTAMP(3);
  if (tid == 0 && blockIdx.x < 65536) for (int i = 0; i < 4; ++i) { csn_gdbg[blockIdx.x * 8 + i] = gst[i]; csn_gdbg[blockIdx.x * 8 + 4 + i] = gin[i]; }
#endif
}

// ---- 256 x 256 x 32 tiles, 8 waves -----------------------------------------------------------------------------------
// Why a second kernel: on this machine the price of a slab is not only its matrix instructions — every 1 KiB wave load
// costs the issuing wave ~90 cycles, and every staged fp32 word a few VALU — and both scale with the BYTES staged per
// FLOP.  A 256 x 256 tile stages half the bytes per FLOP of a 128 x 128 one (64 KB for 4.2 MFLOP), at the same 8 loads per
// thread and slab.  Waves: 4 (M) x 2 (N), 64 x 128 accumulators each (128 registers).  LDS: k-contiguous planes are not
// padded but XOR-swizzled (16-byte unit u of row r lives at u ^ ((r >> 2) & 3): conflict-free 16-byte fragment reads),
// 128 KB (136 KB with a k-major B) for two stages.
// LN: the out-projection of the attention layer (csa_models.py:114-118) — M = d_model = 256 is exactly one tile, so the
// epilogue can apply the fc dropout, add the residual and LayerNorm every point over its 256 channels (per-point sums are
// combined across the four M-waves through LDS) before anything is written: C = xhat, q carries the epilogue's operands.
// AF / BF (single-product modes): format of the A / B input (CSN_FMT_*): 16-bit activation maps are staged by copy — 16-byte
// loads of 8 elements, half the loads and none of the conversion work of the fp32 form (CSN_FMT_F16_TO_BF16: converted in
// registers).  Contraction counts are multiples of 4: the last 16-byte unit of a k-contiguous row may be half valid, its
// upper half is then cleared (what lies behind it is the next block's data, not zeros).
template <typename PR, bool B_NK, bool BT, bool LN = false, int AF = 0, int BF = 0>
__global__ __launch_bounds__(512, 2) void csn_gemm_bf16x3_big_kernel(CsnGemmArgs p, CsnOutProjArgs q) {
  static_assert(!BT || !B_NK, "tile-plane B is k-major");
  static_assert(!LN || (!B_NK && !BT), "LayerNorm epilogue: W_fc (MK) x Ctx^T (KN)");
  static_assert((AF == 0 && BF == 0) || PR::NPL == 1, "16-bit operand maps: single-product modes only");
  static_assert(!(BT && BF) && !((AF == 2 || BF == 2) && PR::HALF), "tile planes carry their own format; fp16 -> bf16 feeds bf16 products");
  constexpr int NPL = PR::NPL;
  constexpr int AES = AF ? 2 : 4, BES = BF ? 2 : 4;     // bytes per element of the A / B (non-tile-plane) input
  constexpr int BM = 256, BN = 256, MT = 2, NT = 4;
  constexpr int PN = BN + 32;                           // pitch of the k-major B planes
  constexpr int A_EL = BM * BK, B_EL = B_NK ? BN * BK : BK * PN;
  // two planes: 128 KB (136 KB with a k-major B); one plane: half of that, but never less than the 128 KB the epilogues
  // use as per-wave fp32 transpose blocks (8 x 16 KB) and the LayerNorm epilogue as its reduction / partial-sum scratch
  constexpr int A_ST = NPL == 2 ? 2 * A_EL : 16384, B_ST = NPL == 2 ? 2 * B_EL : (B_EL > 16384 ? B_EL : 16384);
  __shared__ __attribute__((aligned(16))) short As_raw[2 * A_ST];   // [stage][plane][row][k ^ swizzle]
  __shared__ __attribute__((aligned(16))) short Bs_raw[2 * B_ST];   // NK: like A   KN: [stage][plane][k][col]
  auto As = [&](int st, int pl) -> short* { return As_raw + st * A_ST + pl * A_EL; };
  auto Bs = [&](int st, int pl) -> short* { return Bs_raw + st * B_ST + pl * B_EL; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  BSTAMP(0);
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 128;

  const int tiles_n = (p.N + BN - 1) / BN, tiles = tiles_n * ((p.M + BM - 1) / BM);
  const int L = blockIdx.x, jj = L >> 3;
  const int tile = jj % tiles;
  int z = (jj / tiles) * 8 + (L & 7);                   // XCD-aware order, as in the kernel above
  if (z >= p.batch) return;
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  const int z0 = z % p.n0; z /= p.n0;
  const int z1 = z % p.n1;
  const int zz = z / p.n1;
  const int it0 = p.grp_off ? p.grp_off[zz] : zz, it1 = p.grp_off ? p.grp_off[zz + 1] : zz + 1;   // items contracted by this tile
  auto item_id = [&](int it) { return p.grp_off ? p.grp_items[it] : (p.eval_ids ? p.eval_ids[it] : it); };
  const int z2 = item_id(it0);
  const int lda = p.A.ld, ldb = p.B.ld, ldc = p.C.ld;
  const int M = p.M, N = (p.n_last > 0 && z0 == p.n0 - 1) ? p.n_last : (p.n_arr ? ((p.n_arr[z2] + 3) & ~3) : p.N);   // ragged batches: this item's own columns (rounded up to 4) / contraction count
  int K = (p.k_last > 0 && z0 == p.n0 - 1) ? p.k_last : (p.k_arr ? p.k_arr[z2] : p.K);
  if (p.k_chunk > 0) { K = min(p.k_chunk, p.K - z0 * p.k_chunk); }
  if (tile_n * BN >= N) return;                                // (a tile beyond a short item's columns; before any barrier)                                // (a tile beyond a short item's columns; before any barrier)
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const bool c_pl = p.C.planes != 0;
  const bool c_tiles = p.C.planes == 2;
  const int c_es = c_pl ? 2 : 4;
  csn_rsrc_t Ar, Br;
  auto set_item = [&](int zi) {
    const char* a_base = reinterpret_cast<const char*>(p.A.ptr) +
                         (p.A.s0 * z0 + p.A.s1 * z1 + p.A.s2 * (long long)(p.A.idx2 ? p.A.idx2[zi] : zi) + (long long)m0 * lda) * AES;
    const long long b_el = p.B.s0 * z0 + p.B.s1 * z1 + p.B.s2 * (long long)(p.B.idx2 ? p.B.idx2[zi] : zi);
    const char* b_base = BT ? reinterpret_cast<const char*>(reinterpret_cast<const short*>(p.B.ptr) + b_el)
                            : reinterpret_cast<const char*>(p.B.ptr) + (b_el + (B_NK ? (long long)n0 * ldb : (long long)n0)) * BES;
    // (16-bit k-contiguous maps: the window ends with the last valid row's K elements — a 16-byte unit may straddle the end
    //  of the contraction, and behind the last row of the last item there is no memory; dwords beyond the window read as 0)
    Ar = csn_make_rsrc(a_base, AF ? csn_kwin_bytes(min(BM, M - m0), lda, K, AES) : (long long)BM * lda * AES);
    Br = csn_make_rsrc(b_base, BT ? (long long)K * ldb * 2
                                  : (B_NK ? (BF ? csn_kwin_bytes(min(BN, N - n0), ldb, K, BES) : (long long)BN * ldb * BES)
                                          : ((long long)(K - 1) * ldb + (N - n0)) * BES));
  };
  set_item(z2);
  char* c_base = reinterpret_cast<char*>(p.C.ptr) + (p.C.s0 * z0 + p.C.s1 * z1 + p.C.s2 * (long long)(p.C.idx2 ? p.C.idx2[z2] : z2) + (long long)m0 * ldc + (c_tiles ? 0 : n0)) * c_es;
  const long long c_win = (long long)BM * ldc * c_es;
  const csn_rsrc_t Cr = csn_make_rsrc(c_base, c_win), Crl = csn_make_rsrc(c_base + p.C.plane_stride * 2, c_pl ? c_win : 0);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: 4 A pieces + 4 B pieces of 16 bytes per thread and slab
  const int pr = tid >> 3, pc = (tid & 7) * 4;          // k-contiguous operands: row pr + 64 i, k = pc .. pc + 3
  const int kr = tid >> 6, kc = (tid & 63) * 4;         // k-major B: k row kr + 8 i, columns kc .. kc + 3
  // tile-plane B: k row t_kr of the slab; its 256 columns are 8 tiles of NPL x 64 bytes = 32 NPL contiguous 16-byte units,
  // unit w = t_j + 16 i (t_j = this thread's place among the 16 threads of the row) -> tile w / (4 NPL), plane, 8 keys
  const int t_kr = (tid >> 3) & 31, t_j = (tid & 7) + 8 * (tid >> 8);
  constexpr int BT_PASS = 2 * NPL;
  const int sw_dst = pr * BK + ((((pc >> 3) ^ ((pr >> 2) & 3)) << 3) | (pc & 4));     // rows 64 apart share the swizzle
  // 16-bit maps, 2 pieces of 16 bytes (8 elements) per thread and slab.  k-contiguous: row pr2 + 128 i, unit pu of the row's 4;
  // k-major B: k row kr2 + 16 i, columns 8 ku .. 8 ku + 7
  const int pr2 = tid >> 2, pu = tid & 3;
  const int kr2 = tid >> 5, ku = tid & 31;
  const int sw16 = pr2 * BK + ((pu ^ ((pr2 >> 2) & 3)) << 3);                         // rows 128 apart share the swizzle
  unsigned a_off[4], b_off[4];
  int bt_dst[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (AF) a_off[i] = (i < 2 && (m0 + pr2 + 128 * i) < M) ? (unsigned)((pr2 + 128 * i) * lda + 8 * pu) * 2u : CSN_OOB;
    else a_off[i] = (m0 + pr + 64 * i) < M ? (unsigned)((pr + 64 * i) * lda + pc) * 4u : CSN_OOB;
    if (BF) {
      if (B_NK) b_off[i] = (i < 2 && (n0 + pr2 + 128 * i) < N) ? (unsigned)((pr2 + 128 * i) * ldb + 8 * pu) * 2u : CSN_OOB;
      else b_off[i] = (i < 2 && (n0 + 8 * ku) < N) ? (unsigned)((kr2 + 16 * i) * ldb + 8 * ku) * 2u : CSN_OOB;
      bt_dst[i] = 0;
    } else if (BT) {
      const int w = t_j + 16 * i, tile = w / (4 * NPL), within = w % (4 * NPL), t_u = within & 3;
      b_off[i] = (i < BT_PASS && (n0 + 32 * tile + 8 * t_u) < N)
                     ? (unsigned)(t_kr * ldb) * 2u + (unsigned)(n0 >> 5) * (unsigned)(64 * NPL) + (unsigned)w * 16u : CSN_OOB;
      bt_dst[i] = (within >> 2) * B_EL + t_kr * PN + 32 * tile + 8 * t_u;
    } else if (B_NK) b_off[i] = (n0 + pr + 64 * i) < N ? (unsigned)((pr + 64 * i) * ldb + pc) * 4u : CSN_OOB;
    else b_off[i] = (n0 + kc) < N ? (unsigned)((kr + 8 * i) * ldb + kc) * 4u : CSN_OOB;
  }
  f32x4 ra[4], rb[4];
  auto load_slab = [&](int k0) {
    const unsigned kp = (k0 + pc) < K ? 0u : CSN_OOB;
    const unsigned kp16 = csn_unit_starts_inside(k0 + 8 * pu, K) ? 0u : CSN_OOB;
    if (AF) {
#pragma unroll
      for (int i = 0; i < 2; ++i) ra[i] = csn_bload4(Ar, a_off[i] | kp16, (unsigned)k0 * 2u);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) ra[i] = csn_bload4(Ar, a_off[i] | kp, (unsigned)k0 * 4u);
    }
    if (BF) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (B_NK) rb[i] = csn_bload4(Br, b_off[i] | kp16, (unsigned)k0 * 2u);
        else rb[i] = csn_bload4(Br, b_off[i] | ((k0 + kr2 + 16 * i) < K ? 0u : CSN_OOB), (unsigned)k0 * (unsigned)ldb * 2u);
      }
    } else if (BT) {
      const unsigned kb = (k0 + t_kr) < K ? 0u : CSN_OOB;
#pragma unroll
      for (int i = 0; i < BT_PASS; ++i) rb[i] = csn_bload4(Br, b_off[i] | kb, (unsigned)k0 * (unsigned)ldb * 2u);
    } else if (B_NK) {
#pragma unroll
      for (int i = 0; i < 4; ++i) rb[i] = csn_bload4(Br, b_off[i] | kp, (unsigned)k0 * 4u);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned kq = (k0 + kr + 8 * i) < K ? 0u : CSN_OOB;
        rb[i] = csn_bload4(Br, b_off[i] | kq, (unsigned)k0 * (unsigned)ldb * 4u);
      }
    }
  };
  // a k-contiguous 16-bit unit of slab k0: converted if need be, its upper half cleared when the contraction ends inside it
  auto unit16 = [&](f32x4 v, int fmt, int k0) {
    if (fmt == CSN_FMT_F16_TO_BF16) v = f16x8_to_bf16x8(v);
    if (csn_unit_upper_half_beyond(k0 + 8 * pu, K)) { v[2] = 0.f; v[3] = 0.f; }
    return v;
  };
  auto store_slab = [&](int st, int k0) {
    s16x4 hi, lo;
    if (AF) {
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(As(st, 0) + sw16 + 128 * BK * i) = unit16(ra[i], AF, k0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        split4<PR>(ra[i], hi, lo);
        *reinterpret_cast<s16x4*>(As(st, 0) + sw_dst + 64 * BK * i) = hi;
        if constexpr (NPL == 2) *reinterpret_cast<s16x4*>(As(st, 1) + sw_dst + 64 * BK * i) = lo;
      }
    }
    if (BF) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (B_NK) *reinterpret_cast<f32x4*>(Bs(st, 0) + sw16 + 128 * BK * i) = unit16(rb[i], BF, k0);
        else *reinterpret_cast<f32x4*>(Bs(st, 0) + (kr2 + 16 * i) * PN + 8 * ku) = BF == CSN_FMT_F16_TO_BF16 ? f16x8_to_bf16x8(rb[i]) : rb[i];
      }
      return;
    }
    if (BT) {
#pragma unroll
      for (int i = 0; i < BT_PASS; ++i) *reinterpret_cast<f32x4*>(Bs(st, 0) + bt_dst[i]) = rb[i];
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      split4<PR>(rb[i], hi, lo);
      const int dst = B_NK ? sw_dst + 64 * BK * i : (kr + 8 * i) * PN + kc;
      *reinterpret_cast<s16x4*>(Bs(st, 0) + dst) = hi;
      if constexpr (NPL == 2) *reinterpret_cast<s16x4*>(Bs(st, 1) + dst) = lo;
    }
  };

  // fragment read positions (lane constants): k-contiguous planes, row l31 of a 32-row tile, 16-byte unit (2 s + h) ^ swizzle
  const int fr_sw = (l31 >> 2) & 3;
  const int grp = lane >> 4, gq = (lane >> 2) & 3, gp = lane & 3;
  const int tr_base = (8 * (grp >> 1) + gq) * PN + 16 * (grp & 1) + 4 * gp;

  const int nk = (K + BK - 1) / BK;
  BSTAMP(1);
  for (int it = it0; it < it1; ++it) {
  if (it > it0) {
    __syncthreads();                                    // the previous item's last slab has been read by every wave
    set_item(item_id(it));
  }
  if (nk > 0) { load_slab(0); store_slab(0, 0); }
  if (nk > 1) load_slab(BK);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      s16x8 ah[MT], al[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int o = (wm0 + 32 * i + l31) * BK + (((2 * s + h) ^ fr_sw) << 3);
        ah[i] = *reinterpret_cast<const s16x8*>(As(cur, 0) + o);
        al[i] = *reinterpret_cast<const s16x8*>(As(cur, NPL - 1) + o);           // (one plane: al = ah, unused)
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        s16x8 bh, bl;
        if (B_NK) {
          const int o = (wn0 + 32 * j + l31) * BK + (((2 * s + h) ^ fr_sw) << 3);
          bh = *reinterpret_cast<const s16x8*>(Bs(cur, 0) + o);
          bl = *reinterpret_cast<const s16x8*>(Bs(cur, NPL - 1) + o);
        } else {
          const int o = tr_base + (16 * s) * PN + wn0 + 32 * j;
          typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
          bh = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(Bs(cur, 0) + o)),
                     __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(Bs(cur, 0) + o + 4 * PN)));
          if constexpr (NPL == 2)
            bl = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(Bs(cur, 1) + o)),
                       __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(Bs(cur, 1) + o + 4 * PN)));
          else bl = bh;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i][j] = mma32<PR>(ah[i], al[i], bh, bl, acc[i][j]);
      }
    }
    if (kt + 1 < nk) {
      store_slab(cur ^ 1, (kt + 1) * BK);
      if (kt + 2 < nk) load_slab((kt + 2) * BK);
    }
    __syncthreads();
  }
  }

  BSTAMP(2);
  // per-wave 16 KB of LDS for the epilogues (the tile loop is over; waves 0..3 take As, 4..7 Bs): 32 rows x 128 columns fp32
  float* wbuf = reinterpret_cast<float*>(wave < 4 ? reinterpret_cast<char*>(As_raw) : reinterpret_cast<char*>(Bs_raw))
                + (wave & 3) * 4096;
  const int cc = lane & 31, rsub = lane >> 5;                         // chunk column (4 floats) and row parity of this lane
  const int col = wn0 + 4 * cc, n = n0 + col;
  const bool n_ok = n < N;                                            // N % 4 == 0: a chunk is all in or all out

  if constexpr (LN) {
    // ---- fc dropout, + residual, LayerNorm over the 256 channels of each point, write xhat and rstd ------------------
    // (stays in the accumulator layout: the 16-byte path through the wave's LDS block — residual rows in, transposed reads —
    //  was built three times and measured 5–20 % slower each time: 40 spilled registers and 128 ds_read_b32 per wave cost
    //  more than the 96 memory instructions it saves)
    float* red = reinterpret_cast<float*>(As_raw);                  // [2][4 M-waves][256 points] (the tile loop is over)
    const int wmi = wave >> 1;
    const long long rs = q.res_index ? q.res_index[z2] : z2;
    const csn_rsrc_t Rr = csn_make_rsrc(q.xres + rs * q.xres_shape_stride + n0, ((long long)(BM - 1) * ldc + (N - n0)) * 4);
    const bool drop = q.dropout_p > 0.f;
    const unsigned thr16 = csn_drop_threshold16(q.dropout_p);
    const unsigned salt = drop ? csn_block_salt((unsigned long long)z2, q.seed) : 0u;
    const float keep_scale = drop ? 1.f / (1.f - q.dropout_p) : 1.f;
    float mean[NT], rstd[NT];
    unsigned voff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int nl = wn0 + 32 * j + l31;
      voff[j] = (n0 + nl) < N ? (unsigned)(4 * h * ldc + nl) * 4u : CSN_OOB;      // (of the fp32 residual; xhat: scaled below)
      float s1 = 0.f;
      unsigned hp = 0;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm0 + 32 * i + csn_acc_row(r, 0);         // + 4 h lives in the lane offset
          float v = acc[i][j][r];
          if (drop) {                                               // rows r, r + 1 (r even) are one channel pair: one hash
            if (!(r & 1)) hp = csn_fc_pair(row + 4 * h, (unsigned)ldc, (unsigned)(n0 + nl), salt);
            v = csn_keep16(hp, r & 1, thr16) ? v * keep_scale : 0.f;
          }
          v += csn_bload(Rr, voff[j], (unsigned)row * (unsigned)ldc * 4u);
          acc[i][j][r] = v;
          s1 += v;
        }
      s1 += csn_xhalf(s1);
      if (h == 0) red[wmi * 256 + nl] = s1;
    }
    __syncthreads();
    BSTAMP(3);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int nl = wn0 + 32 * j + l31;
      mean[j] = (red[nl] + red[256 + nl] + red[512 + nl] + red[768 + nl]) * (1.f / BM);
      float s2 = 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float dl = acc[i][j][r] - mean[j];
          s2 += dl * dl;
        }
      s2 += csn_xhalf(s2);
      if (h == 0) red[1024 + wmi * 256 + nl] = s2;
    }
    __syncthreads();
    BSTAMP(4);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int nl = wn0 + 32 * j + l31;
      rstd[j] = 1.f / sqrtf((red[1024 + nl] + red[1280 + nl] + red[1536 + nl] + red[1792 + nl]) * (1.f / BM) + q.eps);
      const bool pt_ok = (n0 + nl) < N;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float xh = (acc[i][j][r] - mean[j]) * rstd[j];
          // xhat as fp32, or (16-bit activation maps) as fp16: |xhat| < sqrt(C), 11 significant bits
          if (c_pl) csn_bstore16(to16<true>(xh), Cr, voff[j] == CSN_OOB ? CSN_OOB : voff[j] >> 1, (unsigned)(wm0 + 32 * i + csn_acc_row(r, 0)) * (unsigned)ldc * 2u);
          else csn_bstore(xh, Cr, voff[j], (unsigned)(wm0 + 32 * i + csn_acc_row(r, 0)) * (unsigned)ldc * 4u);
          acc[i][j][r] = pt_ok ? xh : 0.f;
        }
      if (wmi == 0 && h == 0 && pt_ok) q.rstd[(long long)z2 * q.n_points + n0 + nl] = rstd[j];
    }
    if (q.sum_ws) {
      // per-channel sums of xhat over this tile's points (the pooled descriptor needs mean_n xhat): lane partials over the
      // wave's 4 column tiles go to LDS as psum[row][wn][32 lanes]; thread (row, wn) adds its 32 (conflict-free rotation),
      // the pair is combined and one float per row leaves for sum_ws[e][tile_n][row]
      float* psum = reinterpret_cast<float*>(Bs_raw);               // 64 KB; `red` lives in As
      const int wn = wave & 1;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm0 + 32 * i + csn_acc_row(r, h);
          psum[(row * 2 + wn) * 32 + l31] = (acc[i][0][r] + acc[i][1][r]) + (acc[i][2][r] + acc[i][3][r]);
        }
      __syncthreads();
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k) s += psum[tid * 32 + ((k + tid) & 31)];
      s += __shfl_xor(s, 1, 64);
      if ((tid & 1) == 0) q.sum_ws[((long long)z2 * tiles_n + tile_n) * 256 + (tid >> 1)] = s;
    }
    BSTAMP(5);
    return;
  }

  // ---- epilogue --------------------------------------------------------------------------------------------------
  // A wave-level memory instruction costs the issuing wave ~100 cycles whatever its width, and an accumulator layout store
  // moves 4 bytes per lane: 128 of them per wave were a third of a short-K tile's time.  So every wave transposes its
  // 64 x 128 block through its own 16 KB of LDS (32 rows at a time; the tile loop is over, no cross-wave traffic) and
  // writes 16 contiguous bytes per lane: 32 stores (and 32 loads when accumulating) instead of 128.
  const float alpha = p.alpha;
  unsigned tcol = 0;
  int tpad = 0;                                                        // 4-key groups of zero padding behind this lane's points
  if (c_tiles) {
    const int Tb = (int)p.C.plane_stride;
    const int blk = n / Tb, kib = n - blk * Tb;                        // Tb % 4 == 0: the 4 points share block and tile
    tcol = (unsigned)(blk * (512 * NPL) + (kib >> 5) * (32 * NPL) + (kib & 31));
    if (n_ok && (kib + 4 == Tb || n + 4 == N)) tpad = ((32 - ((kib + 4) & 31)) & 31) >> 2;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r] * alpha;
        if ((m0 + wm0 + 32 * i + csn_acc_row(r, h)) < p.div_rows) v = v / p.div_val;
        wbuf[csn_acc_row(r, h) * 128 + 32 * j + l31] = v;
      }
    f32x4 vals[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) vals[t] = *reinterpret_cast<const f32x4*>(&wbuf[(2 * t + rsub) * 128 + 4 * cc]);
    unsigned off[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int ml = wm0 + 32 * i + 2 * t + rsub;
      const bool ok = n_ok && (m0 + ml) < M;
      off[t] = !ok ? CSN_OOB : (c_tiles ? ((unsigned)(ml * ldc) + tcol) * 2u : (unsigned)(ml * ldc + col) * (unsigned)c_es);
    }
    if (c_pl) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        s16x4 hi, lo;
        split4<PR>(vals[t], hi, lo);
        const u32x2 h2 = __builtin_bit_cast(u32x2, hi), l2 = __builtin_bit_cast(u32x2, lo);
        __builtin_amdgcn_raw_buffer_store_b64(h2, Cr, off[t], 0, 0);
        if constexpr (NPL == 2) {
          if (c_tiles) __builtin_amdgcn_raw_buffer_store_b64(l2, Cr, off[t], 64, 0);
          else __builtin_amdgcn_raw_buffer_store_b64(l2, Crl, off[t], 0, 0);
        }
      }
      if (c_tiles && tpad > 0) {
        // this lane's 4 points end their block (or the map): the keys up to the end of that 32-key tile are padding the
        // attention kernels read as zeros — written here, so that no caller has to clear them
        const u32x2 z2 = {0u, 0u};
#pragma unroll
        for (int t = 0; t < 16; ++t)
          for (int g = 1; g <= tpad; ++g) {
            const unsigned po = off[t] == CSN_OOB ? CSN_OOB : off[t] + 8u * (unsigned)g;
            __builtin_amdgcn_raw_buffer_store_b64(z2, Cr, po, 0, 0);
            if constexpr (NPL == 2) __builtin_amdgcn_raw_buffer_store_b64(z2, Cr, po, 64, 0);
          }
      }
    } else {
      if (p.accumulate) {
        f32x4 prev[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) prev[t] = csn_bload4(Cr, off[t]);
#pragma unroll
        for (int t = 0; t < 16; ++t) vals[t] += prev[t];
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) csn_bstore4(vals[t], Cr, off[t]);
    }
  }
  BSTAMP(5);
}

// ---- 256 x 256 x 32 tiles on SIXTEEN waves ---------------------------------------------------------------------------
// The same tile as the kernel above with twice the waves: 4 (M) x 4 (N) waves of 64 x 64 accumulators (64 registers instead
// of 128), 1024 threads, two staging pieces per thread and operand instead of four — four waves per SIMD if a wave stays within
// 128 registers.  A fp32, k-contiguous; B k-major (fp32, a 16-bit map, or tile planes with the grouped item loop) or
// k-contiguous fp32; C fp32 (optionally accumulated into) or a 16-bit map
// through the per-wave LDS transpose).  Measured (scripts/dev/wide_gemm_probe.py, same bits as the 8-wave kernel): the
// projection of 128 slots 0.79 -> 0.71 ms in bf16x3, 0.61 -> 0.57 ms in bf16.  It serves the bf16x3 mode: the plain products
// (Q, dCtx), the grouped tile-plane products dV / dK (B_NK = false, BT) and the weight gradients (B_NK) — config-3 step 28.09 ->
// 27.70 ms, gradients bit for bit those of the 8-wave kernel (scripts/dev/wide_ab_step.py); the LayerNorm epilogue and the
// 16-bit operand forms of the one-plane modes stay on the 8-wave kernel (measured there: +-0).
template <typename PR, bool B_NK, bool BT, int BF>
__global__ __launch_bounds__(1024, 4) void csn_gemm_bf16x3_wide_kernel(CsnGemmArgs p) {
  static_assert(BF == 0 || (PR::NPL == 1 && !B_NK && !BT), "16-bit operand maps: single-product modes, k-major B");
  static_assert(!BT || !B_NK, "tile-plane B is k-major");
  constexpr int NPL = PR::NPL;
  constexpr int BM = 256, BN = 256, MT = 2, NT = 2;
  constexpr int PN = BN + 32;
  constexpr int A_EL = BM * BK, B_EL = B_NK ? BN * BK : BK * PN;
  constexpr int BES = BF ? 2 : 4;
  // (each array also serves 8 waves as their 8 KB epilogue blocks: never less than 64 KB)
  constexpr int A_ALL = 2 * NPL * A_EL > 32768 ? 2 * NPL * A_EL : 32768, B_ALL = 2 * NPL * B_EL > 32768 ? 2 * NPL * B_EL : 32768;
  __shared__ __attribute__((aligned(16))) short As_raw[A_ALL];   // [stage][plane][row][k ^ swizzle]
  __shared__ __attribute__((aligned(16))) short Bs_raw[B_ALL];   // NK: like A   KN: [stage][plane][k][col]
  auto As = [&](int st, int pl) -> short* { return As_raw + (st * NPL + pl) * A_EL; };
  auto Bs = [&](int st, int pl) -> short* { return Bs_raw + (st * NPL + pl) * B_EL; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm0 = (wave >> 2) * 64, wn0 = (wave & 3) * 64;
  const int tiles_n = (p.N + BN - 1) / BN, tiles = tiles_n * ((p.M + BM - 1) / BM);
  const int L = blockIdx.x, jj = L >> 3;
  const int tile = jj % tiles;
  int z = (jj / tiles) * 8 + (L & 7);
  if (z >= p.batch) return;
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  const int z0 = z % p.n0; z /= p.n0;
  const int z1 = z % p.n1;
  const int zz = z / p.n1;
  const int it0 = p.grp_off ? p.grp_off[zz] : zz, it1 = p.grp_off ? p.grp_off[zz + 1] : zz + 1;   // items contracted by this tile
  auto item_id = [&](int it) { return p.grp_off ? p.grp_items[it] : (p.eval_ids ? p.eval_ids[it] : it); };
  const int z2 = item_id(it0);
  const int lda = p.A.ld, ldb = p.B.ld, ldc = p.C.ld;
  const int M = p.M, N = (p.n_last > 0 && z0 == p.n0 - 1) ? p.n_last : (p.n_arr ? ((p.n_arr[z2] + 3) & ~3) : p.N);
  int K = (p.k_last > 0 && z0 == p.n0 - 1) ? p.k_last : (p.k_arr ? p.k_arr[z2] : p.K);
  if (p.k_chunk > 0) { K = min(p.k_chunk, p.K - z0 * p.k_chunk); }
  if (tile_n * BN >= N) return;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const bool c_pl = p.C.planes != 0;
  const int c_es = c_pl ? 2 : 4;
  csn_rsrc_t Ar, Br;
  auto set_item = [&](int zi) {
    const float* a_base = p.A.ptr + p.A.s0 * z0 + p.A.s1 * z1 + p.A.s2 * (long long)(p.A.idx2 ? p.A.idx2[zi] : zi) + (long long)m0 * lda;
    const long long b_el = p.B.s0 * z0 + p.B.s1 * z1 + p.B.s2 * (long long)(p.B.idx2 ? p.B.idx2[zi] : zi);
    const char* b_base = BT ? reinterpret_cast<const char*>(reinterpret_cast<const short*>(p.B.ptr) + b_el)
                            : reinterpret_cast<const char*>(p.B.ptr) + (b_el + (B_NK ? (long long)n0 * ldb : (long long)n0)) * BES;
    Ar = csn_make_rsrc(a_base, (long long)BM * lda * 4);
    Br = csn_make_rsrc(b_base, BT ? (long long)(p.B.planes == 3 ? p.K : K) * ldb * 2
                                  : (B_NK ? (long long)BN * ldb * BES : ((long long)(K - 1) * ldb + (N - n0)) * BES));
  };
  set_item(z2);
  char* c_base = reinterpret_cast<char*>(p.C.ptr) + (p.C.s0 * z0 + p.C.s1 * z1 + p.C.s2 * (long long)(p.C.idx2 ? p.C.idx2[z2] : z2) + (long long)m0 * ldc + n0) * c_es;
  const csn_rsrc_t Cw = csn_make_rsrc(c_base, (long long)min(BM, M - m0) * ldc * c_es);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging, two 16-byte pieces per thread, operand and slab: k-contiguous operands row pr + 128 i, k = pc .. pc + 3; k-major
  // fp32 B k row kr + 16 i, columns kc .. kc + 3; a 16-bit B map ONE piece (k row kr2, columns 8 ku .. + 7); tile-plane B: k row
  // t_kr, its 256 columns are 32 NPL contiguous units, unit w = t_j + 32 i -> tile w / (4 NPL), plane, 8 keys
  const int pr = tid >> 3, pc = (tid & 7) * 4;
  const int kr = tid >> 6, kc = (tid & 63) * 4;
  const int kr2 = tid >> 5, ku = tid & 31;
  const int t_kr = tid >> 5, t_j = tid & 31;
  const int sw_dst = pr * BK + ((((pc >> 3) ^ ((pr >> 2) & 3)) << 3) | (pc & 4));     // rows 128 apart share the swizzle
  unsigned a_off[2], b_off[2];
  int bt_dst[2], bt_row[2];
  const bool b_tm = BT && p.B.planes == 3;              // tile-major tile planes (CsnAttnArgs::sc_layout)
  const unsigned bt_kstep = b_tm ? (unsigned)(64 * NPL) : (unsigned)ldb * 2u;      // bytes from one k row to the next
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    a_off[i] = (m0 + pr + 128 * i) < M ? (unsigned)((pr + 128 * i) * lda + pc) * 4u : CSN_OOB;
    bt_dst[i] = 0; bt_row[i] = 0;
    if (BT) {
      if (b_tm) {
        // tile-major planes: per block [key tile][k row = query][hi 32 | lo 32].  Unit u = tid + 1024 i of the slab's 32 rows x 8
        // tiles x 4 NPL units: within the row's tile (u % (4 NPL)), row (next 5 bits), tile — a wave reads 8 rows of one tile,
        // 1 KB in one piece
        const int u = tid + 1024 * i, within = u % (4 * NPL), t_u = within & 3, q = (u / (4 * NPL)) & 31, tl = u / (128 * NPL);
        b_off[i] = (i < NPL && (n0 + 32 * tl + 8 * t_u) < N)
                       ? (unsigned)(((n0 >> 5) + tl) * p.K + q) * (unsigned)(64 * NPL) + (unsigned)within * 16u : CSN_OOB;
        bt_dst[i] = (within >> 2) * B_EL + q * PN + 32 * tl + 8 * t_u;
        bt_row[i] = q;
      } else {
        const int w = t_j + 32 * i, tl = w / (4 * NPL), within = w % (4 * NPL), t_u = within & 3;
        b_off[i] = (i < NPL && (n0 + 32 * tl + 8 * t_u) < N)
                       ? (unsigned)(t_kr * ldb) * 2u + (unsigned)(n0 >> 5) * (unsigned)(64 * NPL) + (unsigned)w * 16u : CSN_OOB;
        bt_dst[i] = (within >> 2) * B_EL + t_kr * PN + 32 * tl + 8 * t_u;
        bt_row[i] = t_kr;
      }
    } else if (BF) b_off[i] = (i == 0 && (n0 + 8 * ku) < N) ? (unsigned)(kr2 * ldb + 8 * ku) * 2u : CSN_OOB;
    else if (B_NK) b_off[i] = (n0 + pr + 128 * i) < N ? (unsigned)((pr + 128 * i) * ldb + pc) * 4u : CSN_OOB;
    else b_off[i] = (n0 + kc) < N ? (unsigned)((kr + 16 * i) * ldb + kc) * 4u : CSN_OOB;
  }
  f32x4 ra[2], rb[2];
  auto load_slab = [&](int k0) {
    const unsigned kp = (k0 + pc) < K ? 0u : CSN_OOB;
#pragma unroll
    for (int i = 0; i < 2; ++i) ra[i] = csn_bload4(Ar, a_off[i] | kp, (unsigned)k0 * 4u);
    if (BT) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) rb[i] = csn_bload4(Br, b_off[i] | ((k0 + bt_row[i]) < K ? 0u : CSN_OOB), (unsigned)k0 * bt_kstep);
    } else if (BF) rb[0] = csn_bload4(Br, b_off[0] | ((k0 + kr2) < K ? 0u : CSN_OOB), (unsigned)k0 * (unsigned)ldb * 2u);
    else if (B_NK) {
#pragma unroll
      for (int i = 0; i < 2; ++i) rb[i] = csn_bload4(Br, b_off[i] | kp, (unsigned)k0 * 4u);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) rb[i] = csn_bload4(Br, b_off[i] | ((k0 + kr + 16 * i) < K ? 0u : CSN_OOB), (unsigned)k0 * (unsigned)ldb * 4u);
    }
  };
  auto store_slab = [&](int st) {
    s16x4 hi, lo;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      split4<PR>(ra[i], hi, lo);
      *reinterpret_cast<s16x4*>(As(st, 0) + sw_dst + 128 * BK * i) = hi;
      if constexpr (NPL == 2) *reinterpret_cast<s16x4*>(As(st, 1) + sw_dst + 128 * BK * i) = lo;
    }
    if (BT) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) *reinterpret_cast<f32x4*>(Bs(st, 0) + bt_dst[i]) = rb[i];
      return;
    }
    if (BF) { *reinterpret_cast<f32x4*>(Bs(st, 0) + kr2 * PN + 8 * ku) = rb[0]; return; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      split4<PR>(rb[i], hi, lo);
      const int dst = B_NK ? sw_dst + 128 * BK * i : (kr + 16 * i) * PN + kc;
      *reinterpret_cast<s16x4*>(Bs(st, 0) + dst) = hi;
      if constexpr (NPL == 2) *reinterpret_cast<s16x4*>(Bs(st, 1) + dst) = lo;
    }
  };
  const int fr_sw = (l31 >> 2) & 3;
  const int grp = lane >> 4, gq = (lane >> 2) & 3, gp = lane & 3;
  const int tr_base = (8 * (grp >> 1) + gq) * PN + 16 * (grp & 1) + 4 * gp;
  const int nk = (K + BK - 1) / BK;
  for (int it = it0; it < it1; ++it) {
  if (it > it0) {
    __syncthreads();                                    // the previous item's last slab has been read by every wave
    set_item(item_id(it));
  }
  if (nk > 0) { load_slab(0); store_slab(0); }
  if (nk > 1) load_slab(BK);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      s16x8 ah[MT], al[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int o = (wm0 + 32 * i + l31) * BK + (((2 * s + h) ^ fr_sw) << 3);
        ah[i] = *reinterpret_cast<const s16x8*>(As(cur, 0) + o);
        al[i] = *reinterpret_cast<const s16x8*>(As(cur, NPL - 1) + o);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        s16x8 bh, bl;
        if (B_NK) {
          const int o = (wn0 + 32 * j + l31) * BK + (((2 * s + h) ^ fr_sw) << 3);
          bh = *reinterpret_cast<const s16x8*>(Bs(cur, 0) + o);
          bl = *reinterpret_cast<const s16x8*>(Bs(cur, NPL - 1) + o);
        } else {
          const int o = tr_base + (16 * s) * PN + wn0 + 32 * j;
          typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
          bh = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(Bs(cur, 0) + o)),
                     __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(Bs(cur, 0) + o + 4 * PN)));
          if constexpr (NPL == 2)
            bl = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(Bs(cur, 1) + o)),
                       __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(Bs(cur, 1) + o + 4 * PN)));
          else bl = bh;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i][j] = mma32<PR>(ah[i], al[i], bh, bl, acc[i][j]);
      }
    }
    if (kt + 1 < nk) {
      store_slab(cur ^ 1);
      if (kt + 2 < nk) load_slab((kt + 2) * BK);
    }
    __syncthreads();
  }
  }
  // epilogue: every wave transposes its 64 x 64 block through its own 8 KB of LDS, 32 rows at a time (the tile loop is over),
  // and writes 16 contiguous bytes per lane (8 in a 16-bit map): 16 stores instead of 64.  Column halves of rows with
  // (row >> 2) & 1 set are swapped, so that the two rows a store instruction writes (r and r + 4) use different banks.
  float* wbuf = reinterpret_cast<float*>(wave < 8 ? reinterpret_cast<char*>(As_raw) : reinterpret_cast<char*>(Bs_raw)) + (wave & 7) * 2048;
  const float alpha = p.alpha;
  const int cc = lane & 15, rsub = lane >> 4;
  const int col = wn0 + 4 * cc;
  const bool n_ok = (n0 + col) < N;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r] * alpha;
        const int row = csn_acc_row(r, h);
        if ((m0 + wm0 + 32 * i + row) < p.div_rows) v = v / p.div_val;
        wbuf[row * 64 + ((32 * j + l31) ^ (32 * ((row >> 2) & 1)))] = v;
      }
    f32x4 vals[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int row = 4 * t + rsub;
      vals[t] = *reinterpret_cast<const f32x4*>(&wbuf[row * 64 + ((4 * cc) ^ (32 * ((row >> 2) & 1)))]);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int ml = wm0 + 32 * i + 4 * t + rsub;
      const unsigned off = (n_ok && (m0 + ml) < M) ? (unsigned)(ml * ldc + col) * (unsigned)c_es : CSN_OOB;
      if (c_pl) {
        s16x4 hi, lo;
        split4<PR>(vals[t], hi, lo);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), Cw, off, 0, 0);
      } else {
        if (p.accumulate) vals[t] += csn_bload4(Cw, off);
        csn_bstore4(vals[t], Cw, off);
      }
    }
  }
}

template <typename PR, bool B_NK, bool BT = false, bool LN = false, int AF = 0, int BF = 0>
int launch_big(const CsnGemmArgs& a, int batch, hipStream_t st, const CsnOutProjArgs* ln = nullptr) {
  CsnGemmArgs b = a;
  b.batch = batch;
  const long long tiles = (long long)((a.N + 255) / 256) * ((a.M + 255) / 256);
  dim3 grid((unsigned)(((batch + 7) / 8) * 8 * tiles));
  CsnOutProjArgs q{};
  if (ln) q = *ln;
  hipLaunchKernelGGL((csn_gemm_bf16x3_big_kernel<PR, B_NK, BT, LN, AF, BF>), grid, dim3(512), 0, st, b, q);
  return (int)hipGetLastError();
}

template <typename PR, int BM, int BN, bool B_NK, bool BT = false, int AF = 0, int BF = 0>
int launch(const CsnGemmArgs& a, int batch, hipStream_t st) {
  CsnGemmArgs b = a;
  b.batch = batch;
  const long long tiles = (long long)((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM);
  dim3 grid((unsigned)(((batch + 7) / 8) * 8 * tiles));
  hipLaunchKernelGGL((csn_gemm_bf16x3_kernel<PR, BM, BN, B_NK, BT, AF, BF>), grid, dim3(256), 0, st, b);
  return (int)hipGetLastError();
}

}  // namespace

// development switches (csn_dev_set, include/csn_hip.h "DEVELOPMENT SECTION")
int csn_gemm_big_tiles = 1;
int csn_gemm_wide = 1;           // plain 256 x 256 products of the bf16x3 mode on the 16-wave kernel (0 off, 2 = the one-plane modes too)
int csn_gemm_wide_set = 7;       // which product forms take it: 1 plain k-major B, 2 tile-plane B (grouped dV / dK), 4 k-contiguous B (weight gradients)

namespace {
template <typename PR, bool B_NK, bool BT, int BF>
int launch_wide(const CsnGemmArgs& a, int batch, hipStream_t st) {
  CsnGemmArgs b = a;
  b.batch = batch;
  const long long tiles = (long long)((a.N + 255) / 256) * ((a.M + 255) / 256);
  dim3 grid((unsigned)(((batch + 7) / 8) * 8 * tiles));
  hipLaunchKernelGGL((csn_gemm_bf16x3_wide_kernel<PR, B_NK, BT, BF>), grid, dim3(1024), 0, st, b);
  return (int)hipGetLastError();
}
}  // namespace

// out-projection + residual + LayerNorm for d_model = 256 on the 256 x 256 tiles (math modes 1..3): xhat = LN(W_fc Ctx^T (+drop) + x)
int csn_launch_outproj_ln_big(const CsnOutProjArgs& a, int mode, hipStream_t st) {
  if (a.C != 256 || (a.D & 3) || (a.ld & 3) || (a.n_points & 3)) return -5;
  // bf16x3 with fp32 maps and D = 256: the weight-stationary streaming kernel (wx_stream.hip; CSN_DEV_WX bit 2 switches it off)
  if (mode == 1 && !a.act16 && !(csn_dev_wx & 4) && csn_wx_takes(256, a.D)) {
    CsnWxArgs w{};
    w.w = a.wfc;
    w.x = a.ctx; w.x_item_stride = a.ctx_eval_stride; w.ldx = a.ld;
    w.out = a.xhat; w.out_item_stride = a.xhat_eval_stride; w.ldo = a.ld;
    w.n_items = a.E; w.n_points = a.n_points; w.n_sets = 1;
    w.div_rows = 0; w.div_val = 1.f; w.div_rcp = 1.f; w.div_exact = 1; w.tb = 0;
    w.res = a.xres; w.res_shape_stride = a.xres_shape_stride; w.res_index = a.res_index;
    w.rstd = a.rstd; w.eps = a.eps; w.dropout_p = a.dropout_p; w.seed = a.seed;
    const int slots = csn_wx_ln_sum_slots(a.E, a.n_points);
    const bool fused = a.xhat_sum && a.sum_ws && a.sum_ws_floats >= (long long)a.E * slots * 256;
    if (fused) { w.sum_ws = a.sum_ws; w.sum_slots = slots; }
    const int rc = csn_launch_wx(w, 3, st);
    if (rc != CSN_NOT_TAKEN) {                           // (a geometry the streaming kernel does not take: the tiled kernel below)
      if (rc || !a.xhat_sum) return rc;
      if (fused) return csn_launch_wx_ln_sums(a.sum_ws, a.xhat_sum, a.E, a.n_points, st);
      return csn_launch_rowsum_f32(a.xhat, a.xhat_sum, (long long)a.E * a.C, a.n_points, a.ld, st, 0);
    }
  }
  CsnGemmArgs g;
  g.A = CsnOperand{const_cast<float*>(a.wfc), 0, 0, 0, nullptr, a.D, 0, 0};
  g.B = CsnOperand{const_cast<float*>(a.ctx), 0, 0, a.ctx_eval_stride, nullptr, a.ld, 0, 0};
  g.C = CsnOperand{a.xhat, 0, 0, a.xhat_eval_stride, nullptr, a.ld, 0, 0};
  if (a.act16) {                                        // Ctx^T arrives as a 16-bit map of the mode's type, xhat leaves as fp16
    if (mode < 2) return -1;
    g.B.fmt = CSN_FMT_16;
    g.C.planes = 1;
  }
  g.M = 256; g.N = a.n_points; g.K = a.D;
  g.n0 = 1; g.n1 = 1; g.k_chunk = 0;
  g.alpha = 1.f; g.div_rows = 0; g.div_val = 1.f; g.accumulate = 0; g.eval_ids = nullptr;
  const int tiles_n = (a.n_points + 255) / 256;
  CsnOutProjArgs b = a;
  const bool fused_sums = a.xhat_sum && a.sum_ws && a.sum_ws_floats >= (long long)a.E * tiles_n * 256;
  if (!fused_sums) b.sum_ws = nullptr;
  // (the 16-wave form of this kernel was built too — same xhat / rstd bits — and measured slower: its epilogue holds the 64
  //  accumulators beside the residual loads in flight under a 128-register bound, 46-48 spills: step 28.36 -> 29.26 ms)
  int rc = a.act16 ? (mode == 3 ? launch_big<F16, false, false, true, 0, 1>(g, a.E, st, &b)
                                : launch_big<Bf16, false, false, true, 0, 1>(g, a.E, st, &b))
           : mode == 3 ? launch_big<F16, false, false, true>(g, a.E, st, &b)
           : mode == 2 ? launch_big<Bf16, false, false, true>(g, a.E, st, &b)
                       : launch_big<Bf16x3, false, false, true>(g, a.E, st, &b);
  if (rc || !a.xhat_sum) return rc;
  if (fused_sums) return csn_launch_partial_sums_f32(a.sum_ws, a.xhat_sum, a.E, tiles_n, 256, st);
  return csn_launch_rowsum_f32(a.xhat, a.xhat_sum, (long long)a.E * a.C, a.n_points, a.ld, st, a.act16 ? 2 : 0);
}

int csn_gemm_bf16x3_big_tiles(int M, int N) { return (csn_gemm_big_tiles && M >= 192 && N >= 224) ? 1 : 0; }
// 1: a dV / dK product of this shape can take its P / dS operand as TILE-MAJOR tile planes (bf16x3, 16-wave kernel)
int csn_gemm_tile_major_planes(int M, int N) { return (csn_gemm_bf16x3_big_tiles(M, N) && csn_gemm_wide) ? 1 : 0; }

namespace {
// 16-bit activation maps as inputs (single-product modes): the combinations the step uses —
//   A fp32 (weights) x B 16 k-major      out-projection input, dCtx^T = W_fc^T dZ^T, dx = W^T dqkv
//   A 16 x B 16 / fp16->bf16 / fp32, NK  weight gradients (dZ Ctx^T; dqkv x^T)
//   A 16 / fp16->bf16 x tile-plane B     dV^T = dO^T P, dK^T = Qs^T dS
template <typename PR, int AF, int BF>
int launch_fmt(const CsnGemmArgs& a, int b_is_nk, int batch, bool big, hipStream_t st) {
  if (a.B.planes == 2) {
    if (b_is_nk || (a.B.ld & 7)) return -1;
    if constexpr (BF == 0 && AF != 0) return big ? launch_big<PR, false, true, false, AF, 0>(a, batch, st) : launch<PR, 128, 128, false, true, AF, 0>(a, batch, st);
    else return -1;
  }
  if (b_is_nk) {
    if constexpr (AF == 1) {
      if (big) return launch_big<PR, true, false, false, AF, BF>(a, batch, st);
      return a.M <= 64 ? launch<PR, 64, 128, true, false, AF, BF>(a, batch, st) : launch<PR, 128, 128, true, false, AF, BF>(a, batch, st);
    } else return -1;
  }
  if constexpr (AF == 0 && BF != 0) {
    if (big) return launch_big<PR, false, false, false, 0, BF>(a, batch, st);
    return a.M <= 64 ? launch<PR, 64, 128, false, false, 0, BF>(a, batch, st) : launch<PR, 128, 128, false, false, 0, BF>(a, batch, st);
  } else return -1;
}

template <typename PR>
int launch_mode(const CsnGemmArgs& a, int b_is_nk, int batch, hipStream_t st) {
  // tiles of 256 x 256 where the output is big enough to fill them (a ragged last tile wastes at most ~3 % here)
  const bool big = csn_gemm_big_tiles && a.M >= 192 && a.N >= 224;
  if (a.grp_off && !big) return -1;                                 // grouped accumulation: 256 x 256 kernel only
  // the 16-wave form of the 256 x 256 tiles: A fp32, C fp32 (or a 16-bit map in the accumulating-free plain case)
  if (csn_gemm_wide && (PR::NPL == 2 || csn_gemm_wide == 2) && big && a.C.planes != 2 && !a.A.fmt && (a.C.planes == 0 || PR::NPL == 1) &&
      !(a.C.planes && a.accumulate)) {
    if (a.B.planes >= 2) {
      if (!b_is_nk && !(a.B.ld & 7) && ((csn_gemm_wide_set & 2) || a.B.planes == 3)) return launch_wide<PR, false, true, 0>(a, batch, st);
    } else if (b_is_nk) {
      if (a.B.fmt == 0 && !a.grp_off && (csn_gemm_wide_set & 4)) return launch_wide<PR, true, false, 0>(a, batch, st);
    } else if (!a.grp_off && (csn_gemm_wide_set & 1)) {
      if (a.B.fmt == 0) return launch_wide<PR, false, false, 0>(a, batch, st);
      if constexpr (PR::NPL == 1) { if (a.B.fmt == 1) return launch_wide<PR, false, false, 1>(a, batch, st); }
    }
  }
  if (a.B.planes == 3) return -1;                                   // tile-major planes: the 16-wave kernel only (csn_gemm_tile_major_planes)
  if (a.A.fmt || a.B.fmt) {
    if constexpr (PR::NPL == 1) {
      const int af = a.A.fmt, bf = a.B.fmt;
      if ((af & 1) && (a.A.ld & 3)) return -2;
      if constexpr (!PR::HALF) {
        if (af == 1 && bf == 0) return launch_fmt<PR, 1, 0>(a, b_is_nk, batch, big, st);
        if (af == 2 && bf == 0) return launch_fmt<PR, 2, 0>(a, b_is_nk, batch, big, st);
        if (af == 1 && bf == 1) return launch_fmt<PR, 1, 1>(a, b_is_nk, batch, big, st);
        if (af == 1 && bf == 2) return launch_fmt<PR, 1, 2>(a, b_is_nk, batch, big, st);
      }
      if (af == 0 && bf == 1) return launch_fmt<PR, 0, 1>(a, b_is_nk, batch, big, st);
    }
    return -1;
  }
  if (a.B.planes == 2) {                                            // tile-plane B: k-major only
    if (b_is_nk || (a.B.ld & 7)) return -1;
    return big ? launch_big<PR, false, true>(a, batch, st) : launch<PR, 128, 128, false, true>(a, batch, st);
  }
  if (big) return b_is_nk ? launch_big<PR, true>(a, batch, st) : launch_big<PR, false>(a, batch, st);
  if (a.M <= 64) return b_is_nk ? launch<PR, 64, 128, true>(a, batch, st) : launch<PR, 64, 128, false>(a, batch, st);
  return b_is_nk ? launch<PR, 128, 128, true>(a, batch, st) : launch<PR, 128, 128, false>(a, batch, st);
}
}  // namespace

// mode: 1 = bf16x3, 2 = bf16, 3 = fp16 (csn_set_math_mode)
int csn_launch_gemm_bf16x3(const CsnGemmArgs& a, int b_is_nk, int batch, int mode, hipStream_t st) {
  if (a.C.planes && a.accumulate) return -1;
  switch (mode) {
    case 1: return launch_mode<Bf16x3>(a, b_is_nk, batch, st);
    case 2: return launch_mode<Bf16>(a, b_is_nk, batch, st);
    case 3: return launch_mode<F16>(a, b_is_nk, batch, st);
    default: return -1;
  }
}
