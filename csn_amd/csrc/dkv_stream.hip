// dV^T = dO^T P  and  dK^T = Qs^T dS  of the attention backward (autograd of MID-FC/csa_models.py:140-142 w.r.t. v and k) on the
// P / dS tile planes the dQ kernel leaves, in the bf16x3 math mode at d_head = 256, as an OUTPUT-STATIONARY STREAM:
//     out[slot][c][key] = sum over the group's evaluations e and the block's queries q of  A_e[c][q] * planes_e[q][key]
// Per (group of evaluations that share the key / value slot, head, block) the output is 256 channels x <= 512 keys; a launch
// reads the planes (2 KB per query row) and the operand map (1 KB per query) once and writes 256 x T floats per unit: it is a
// byte stream (64 KB per 32-query chunk for 48 matrix instructions a wave), which the 256 x 256-tile kernel of gemm_bf16x3.hip
// ran at 4.05 TB/s with the matrix pipe 49 % busy — one slab in flight, a prologue and a 256 KB epilogue per tile with nothing
// in flight.  Here, in the manner of wx_stream.hip:
//   * persistent grid, one 8-wave work-group per CU; a work-group owns the 256 x 256 accumulator tile (wave w: channel rows
//     32 w .. + 31 as 8 tiles of v_mfma_f32_32x32x16_bf16, 128 registers) of ONE (unit, key half) at a time and walks a fixed
//     list of them; the two halves of a unit sit on work-groups of one XCD (equal blockIdx % 8) and walk the same chunks, so
//     the operand chunk leaves HBM once (speed only);
//   * the stream of 32-query chunks never stops at a unit's end: requests run three chunks (A) / two chunks (planes) ahead of
//     the contraction across units; a unit's end is an epilogue (the tile out through per-wave LDS transposes, 16-byte row
//     stores, accumulators cleared) with the next unit's chunks already in flight;
//   * A (fp32, k-contiguous) is WAVE-PRIVATE — wave w contracts rows 32 w .. + 31 and nobody else reads them: the wave fetches
//     its own 32 rows x 128 bytes as 16-byte row pieces into register sets (requests hidden from the compiler and waited for by
//     hand, wx_common.h), splits them into bf16 hi / lo planes in its own 4 KB of LDS and reads the fragments back — no barrier
//     stands between a wave's A operand and its matrix instructions, and the two waves of a SIMD need not march in step;
//   * the planes need no conversion: every 1 KB row piece [8 tiles of hi 32 | lo 32] goes from memory straight into one of THREE
//     LDS stages by LDS-DMA (no registers, no vector instructions) at a row pitch of 1088 bytes, where the transposing
//     ds_read_b64_tr_b16 of a 32-lane half (4 consecutive rows x 64 bytes) touches every bank once.
// The sums are formed in the tiled kernel's order (evaluations of a group in list order, chunks ascending, two k steps, lo x hi,
// hi x lo, hi x hi): the results are bit for bit those of the GEMM route (tests/test_gpu_dkv_stream.py).
#include "csn_common.h"
#include "csn_kernels.h"
#include "wx_common.h"
#include <type_traits>

namespace {

constexpr int DK_PB = 544;                        // plane-stage row pitch in 16-bit elements: 1 KB of tiles + 64 bytes
constexpr int DK_BST = 32 * DK_PB;                // 16-bit elements of a plane stage (32 query rows)
constexpr int DK_NB = 3;                          // plane stages
constexpr int DK_AST = 2 * WX_PLANE;              // the A stage: hi + lo planes of [256 channels][32 queries]
constexpr int DK_EB = 16 * 32;                    // floats of a wave's epilogue block (16 rows x 32 keys)

template <int N>
CSN_DEVINL void dk_wait(f32x4* R) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]) : "n"(N) : "memory");
}

// place in the work list: unit U = ((g * H) + hd) * n_blocks + b, evaluation list position it of group g, chunk t of the block
struct DkCur { int U, hd, b, it, it0, it1, t, nt, Tb, e, ai; };

__global__ __launch_bounds__(512, 2) void csn_dkv_stream_kernel(CsnDkvStreamArgs p) {
  __shared__ __attribute__((aligned(16))) short smem[DK_AST + DK_NB * DK_BST + 8 * DK_EB * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  short* const As = smem;
  short* const Bs = smem + DK_AST;
  float* const eb = reinterpret_cast<float*>(smem + DK_AST + DK_NB * DK_BST) + wave * DK_EB;
  const unsigned bs_lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((short __attribute__((address_space(3)))*)smem)) + 2u * DK_AST;

  // work list of this work-group: units u0, u0 + ustep, ...; key half `half` of each
  const int nh = p.T > 256 ? 2 : 1;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, P = (int)(gridDim.x >> 3);
  const int half = j % nh, u0 = xcd * (P / nh) + j / nh, ustep = 8 * (P / nh);
  const int n_units = p.n_groups * p.H * p.n_blocks;
  const int key0 = 256 * half;

  auto set_item = [&](DkCur& c) __attribute__((always_inline)) {                      // evaluation of list position c.it and its operand map
    c.e = p.items ? wx_sload(p.items + c.it) : c.it;
    c.ai = p.a_index ? wx_sload(p.a_index + c.e) : c.e;
  };
  auto set_unit = [&](DkCur& c, int U) __attribute__((always_inline)) {               // first chunk of the first unit >= U of this list that has keys in this half
    for (;; U += ustep) {
      if (U >= n_units) { c.U = -1; c.e = 0; c.ai = 0; c.hd = 0; c.b = 0; c.t = 0; c.nt = 1; c.Tb = 0; c.it = c.it0 = c.it1 = 0; return; }
      const int b = U % p.n_blocks, r = U / p.n_blocks;
      const int Tb = (p.T_last > 0 && b == p.n_blocks - 1) ? p.T_last : p.T;
      if (key0 >= Tb) continue;
      const int g = r / p.H;
      c.U = U; c.b = b; c.hd = r % p.H; c.Tb = Tb; c.nt = (Tb + 31) >> 5; c.t = 0;
      c.it0 = p.grp_off ? wx_sload(p.grp_off + g) : g;
      c.it1 = p.grp_off ? wx_sload(p.grp_off + g + 1) : g + 1;
      c.it = c.it0;
      set_item(c);
      return;
    }
  };
  auto advance = [&](DkCur& c) __attribute__((always_inline)) {
    if (c.U < 0) return;
    if (++c.t < c.nt) return;
    c.t = 0;
    if (++c.it < c.it1) { set_item(c); return; }
    set_unit(c, c.U + ustep);
  };

  // ---- A: the wave's own rows 32 wave + lane / 8 + 8 i of the 256 channels, queries 4 (lane % 8) .. + 3 of the chunk ---------
  const int arow = lane >> 3, c4 = lane & 7;
  const unsigned a_voff = (unsigned)((32 * wave + arow) * p.ld + 4 * c4) * 4u;
  auto issueA = [&](const DkCur& c, f32x4* R) __attribute__((always_inline)) {
    const bool ex = c.U >= 0 && !(p.ablate & 1);
    const int valid = ex ? min(32, c.Tb - 32 * c.t) : 0;
    const long long col0 = (long long)c.b * p.T + 32 * c.t;
    const u32x4 Ar = wx_rsrc(p.a + (long long)c.ai * p.a_stride + (long long)(256 * c.hd) * p.ld + col0, ((long long)255 * p.ld + valid) * 4);
    const unsigned off = 4 * c4 < valid ? a_voff : CSN_OOB;
#pragma unroll
    for (int i = 0; i < 4; ++i) wx_request(R[i], Ar, off, (unsigned)(8 * i * p.ld) * 4u);
  };
  // the wave's LDS image of an A plane: [32 rows][32 queries] bf16, 64 bytes a row; 16-byte unit u of row r sits at u ^ ((r >> 2) & 3)
  short* const Aw = As + wave * (2 * 32 * WX_CH);                   // hi plane, then lo plane: 4 KB a wave
  auto commitA = [&](const f32x4* R) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = arow + 8 * i;
      const int dst = row * WX_CH + 8 * ((c4 >> 1) ^ ((row >> 2) & 3)) + 4 * (c4 & 1);
      s16x4 hi, lo;
      split4<Bf16x3>(R[i], hi, lo);
      *reinterpret_cast<s16x4*>(Aw + dst) = hi;
      *reinterpret_cast<s16x4*>(Aw + dst + 32 * WX_CH) = lo;
    }
  };
  // ---- planes: query rows wave + 8 i of the chunk, 1 KB each (this half's 8 tiles), by LDS-DMA ---------------------------
  const long long blk_el = (long long)p.T * p.Tp * 2;                 // 16-bit elements of a block's planes (the bytes of its fp32 scores)
  auto issueB = [&](const DkCur& c, int stage) __attribute__((always_inline)) {
    const bool ex = c.U >= 0 && !(p.ablate & 2);
    const int valid = ex ? min(32, c.Tb - 32 * c.t) : 0;
    const short* base = p.planes + (((long long)c.e * p.H + c.hd) * p.n_blocks + c.b) * blk_el + (long long)(32 * c.t) * (2 * p.Tp);
    const u32x4 Br = wx_rsrc(base, (long long)valid * p.Tp * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = wave + 8 * i;
      wx_dma(bs_lds + (unsigned)(stage * DK_BST + r * DK_PB) * 2u, Br, r < valid ? (unsigned)(16 * lane + 4 * key0) : CSN_OOB, (unsigned)(r * p.Tp) * 4u);
    }
  };

  // ---- contraction of one chunk --------------------------------------------------------------------------------------
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int f_row = l31 * WX_CH, f_sw = (l31 >> 2) & 3;
  auto fragA = [&](const short* plane, int s) __attribute__((always_inline)) {
    return *reinterpret_cast<const s16x8*>(plane + f_row + 8 * ((2 * s + h) ^ f_sw));
  };
  // B fragment of k step s, key tile t: lane l holds planes[16 s + 8 (l >> 5) + j][key 32 t + (l & 31)].  Transposing read as in
  // wx_stream.hip: 16-lane group g covers keys 16 (g & 1) .. + 15 and query rows 8 (g >> 1) .. + 7 in two passes of 4 rows
  const int grp = lane >> 4, gq = (lane >> 2) & 3, gp = lane & 3;
  const int tr_base = (8 * (grp >> 1) + gq) * DK_PB + 16 * (grp & 1) + 4 * gp;
  auto compute = [&](int stage) __attribute__((always_inline)) {
    const short* B = Bs + stage * DK_BST + tr_base;
    if (p.ablate & 4) { acc[0][0] += __builtin_bit_cast(float, (int)B[0]); return; }
    s16x8 ah[2], al[2], bh[2], bl[2];
    auto rd = [&](int i, s16x8& fh, s16x8& fl) __attribute__((always_inline)) {
      const short* a = B + 16 * (i >> 3) * DK_PB + 64 * (i & 7);
      fh = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a)), __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + 4 * DK_PB)));
      fl = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + 32)), __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + 32 + 4 * DK_PB)));
    };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 2; ++s) { ah[s] = fragA(Aw, s); al[s] = fragA(Aw + 32 * WX_CH, s); }
#pragma unroll
    for (int i = 0; i < 2; ++i) rd(i, bh[i], bl[i]);
    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = i & 7, s = i >> 3, r = i & 1;
      acc[t] = wx_mma(ah[s], al[s], bh[r], bl[r], acc[t]);
      if (i + 2 < 16) rd(i + 2, bh[r], bl[r]);
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- a unit's end: the wave's 32 rows x 256 keys out through its LDS block, 16 rows x 32 keys at a time ----------------
  const int erow = lane >> 3, c8 = lane & 7;
  auto epilogue = [&](const DkCur& c) __attribute__((always_inline)) {
    const int e0 = p.items ? wx_sload(p.items + c.it0) : c.it0;
    const long long slot = p.out_index ? (long long)wx_sload(p.out_index + e0) : (long long)e0;
    const int kvalid = (p.ablate & 8) ? 0 : min(256, c.Tb - key0);
    const csn_rsrc_t Or = csn_make_rsrc(p.out + slot * p.out_stride + (long long)(256 * c.hd + 32 * wave) * p.ld + (long long)c.b * p.T + key0,
                                        ((long long)31 * p.ld + kvalid) * 4);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) eb[(csn_acc_row(8 * hf + rr, h) & 15) * 32 + l31] = acc[t][8 * hf + rr];
        const unsigned off = (32 * t + 4 * c8) < kvalid ? (unsigned)((16 * hf + erow) * p.ld + 32 * t + 4 * c8) * 4u : CSN_OOB;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          f32x4 v = *reinterpret_cast<const f32x4*>(&eb[(erow + 8 * k) * 32 + 4 * c8]);
          if (p.accumulate) v += csn_bload4(Or, off, (unsigned)(8 * k * p.ld) * 4u);
          csn_bstore4(v, Or, off, (unsigned)(8 * k * p.ld) * 4u);
        }
      }
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  };

  // ---- the stream ----------------------------------------------------------------------------------------------------
  // Chunk k travels in register set k % 3 (A) and lands in plane stage k % 3; requests run two chunks ahead of the contraction.
  // Iteration c: plane pieces of chunk c + 2 | wait for A of chunk c, commit it to the wave's own block | request A of chunk
  // c + 2 | contract chunk c | wait for the wave's plane pieces of chunk c + 1 | ONE barrier (the planes of chunk c + 1 have landed
  // for every wave, and every wave is done with the stage chunk c + 3 will take).  Behind the A request of chunk c (end of
  // iteration c - 2) the wave has issued 8 + 4 operations, behind the plane pieces of chunk c + 1 (top of iteration c - 1)
  // 4 + 8: both waits count 12 (a unit's epilogue stores come on top and only make a wait longer).  Requests beyond the list's
  // end are made with every lane off, so the counts hold in every iteration.
  DkCur cL, cC;
  set_unit(cC, u0);
  cL = cC;
  if (cC.U < 0) return;
  // plane stages start as zeros: rows and keys that no DMA ever writes meet zero operand values, but must not hold NaN patterns
  for (int i = tid; i < DK_NB * DK_BST / 8; i += 512) reinterpret_cast<f32x4*>(Bs)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  f32x4 R0[4], R1[4], R2[4];
  issueB(cL, 0); issueA(cL, R0); advance(cL);
  issueB(cL, 1); issueA(cL, R1); advance(cL);
  dk_wait<0>(R0);
  dk_wait<0>(R1);
  __syncthreads();
  auto iteration = [&](auto st_c, f32x4* RC, f32x4* RQ) __attribute__((always_inline)) {
    constexpr int ST = decltype(st_c)::value;
    issueB(cL, (ST + 2) % 3);                          // planes of chunk c + 2 (their stage held chunk c - 1)
    dk_wait<12>(RC);
    commitA(RC);                                       // chunk c, into the wave's own block
    issueA(cL, RQ); advance(cL);                       // A of chunk c + 2
    compute(ST);
    if (cC.t == cC.nt - 1 && cC.it == cC.it1 - 1) epilogue(cC);
    advance(cC);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // this wave's plane pieces of chunk c + 1 have landed
    __syncthreads();
  };
  while (true) {
    if (cC.U < 0) break;
    iteration(std::integral_constant<int, 0>{}, R0, R2);
    if (cC.U < 0) break;
    iteration(std::integral_constant<int, 1>{}, R1, R0);
    if (cC.U < 0) break;
    iteration(std::integral_constant<int, 2>{}, R2, R1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (requests beyond the end: every lane off; nothing may land after the exit)
}

}  // namespace

int csn_dev_dkv_stream = 0;      // measured: the GEMM route is as fast over the step (profiles/r5_dkv_stream.txt)

bool csn_dkv_stream_takes(int d_head, int block, int score_pitch) {
  const int nh = block > 256 ? 2 : 1;
  return csn_dev_dkv_stream != 0 && d_head == 256 && block > 0 && block <= 512 && !(block & 3) && score_pitch >= 256 * nh && !(score_pitch & 31) &&
         ((wx_grid() >> 3) % nh) == 0;
}

int csn_launch_dkv_stream(const CsnDkvStreamArgs& a, hipStream_t st) {
  if (a.n_groups <= 0 || a.H <= 0 || a.n_blocks <= 0) return 0;
  if (!csn_dkv_stream_takes(256, a.T, a.Tp) || (a.ld & 3) || (a.a_stride & 3) || (a.out_stride & 3)) return CSN_NOT_TAKEN;
  // 32-bit element offsets inside one map / one block of planes
  if ((long long)a.H * 256 * a.ld >= (1ll << 29) || (long long)a.T * a.Tp >= (1ll << 28)) return CSN_NOT_TAKEN;
  CsnDkvStreamArgs b = a;
  b.ablate = csn_dev_dkv_stream >> 4;                  // development: timing-only ablations (bits 4..7 of CSN_DEV_DKV_STREAM)
  hipLaunchKernelGGL(csn_dkv_stream_kernel, dim3(wx_grid()), dim3(512), 0, st, b);
  return (int)hipGetLastError();
}
