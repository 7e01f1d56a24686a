// Fused block attention FORWARD for the reference's own geometry (d_k = d_v = 256, MID-FC/csa_models.py:138-144, 147) in the
// bf16x3 mode, built around v_mfma_f32_32x32x16_bf16: FOUR waves of 32 queries, one wave per SIMD, the whole 512-register file.
//
// Why a second forward beside attn_bf16x3.hip.  That kernel (8 waves x 16 queries on v_mfma_f32_16x16x32_bf16) runs its matrix
// pipe 40 % busy and its parts add up instead of overlapping: a 16x16x32 instruction holds the SIMD's vector issue for 8 of
// its 16 cycles, so with two waves per SIMD the softmax / dropout / hi-lo split of one wave runs at half speed beside the
// partner's matrix phase, and every wave re-reads the whole K and V tile from LDS for only 16 queries.  A 32x32x16 instruction
// holds the vector issue for 8 of 32 cycles and feeds 32 queries per fragment read: half the LDS bytes per FLOP, three quarters
// of the vector issue slots free while the pipe runs.  It needs N = 32 query columns per wave, i.e. Q (128 registers as hi / lo
// B fragments) and O (128 accumulator registers) of 32 queries in ONE wave: more than the 256 registers of a wave at two per
// SIMD — hence one wave per SIMD, and everything a partner wave used to hide has to be hidden inside the wave's own stream:
//   * software pipeline over the key tiles: iteration t issues the 48 matrix instructions of S(t+1) = K(t+1) Qs^T beside the
//     pointwise work (exp2, row sums, dropout, hi / lo split) of tile t, then the 48 of O += V(t) P(t);
//   * P never leaves registers: the S accumulators of a lane are, after the split, the B fragments of the second product as
//     they stand (the V fragment is read in the matching key order: keys 16 s + 4 h .. + 3 and 16 s + 8 + 4 h .. + 3);
//   * K / V tiles travel global -> registers -> LDS one iteration ahead (requests by inline asm, waited for by hand: the
//     score stores of the loop would otherwise drain them, see wx_stream.hip), ONE barrier per tile.
// Same arithmetic as attn_bf16x3.hip (three products per FLOP, small terms first, exp2 softmax with the lazy re-basing, the same
// counter-based dropout masks, the same lse) in another summation order of the d index: results agree to fp32 rounding.
// Scope: tile-plane K / V, block mode, fp32 Qs in / fp32 Ctx out, row-major scores; everything else stays on attn_bf16x3.hip.
#include "csn_common.h"
#include "csn_kernels.h"

// -DCSN_X4_STAMPS: diagnostic build that records s_memtime at the part boundaries of tiles 4..11 of the first work-groups
// (scripts/x4_stamps.py); no stamp executes in the product build
#ifdef CSN_X4_STAMPS
__device__ unsigned long long csn_x4_dbg[256 * 4 * 8 * 8];
extern "C" int csn_x4_debug_read(void* dst, long long bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(csn_x4_dbg), bytes); }
#define XSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (dbg_on) stamps[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define XSTAMP(i)
#endif

namespace {

using namespace csn_mode;
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;

constexpr int XD = 256, XKT = 32;
constexpr int XPLANE = XD * XKT;               // 16-bit elements of one plane of one tile image (16 KB)
constexpr int XIMG = 2 * XPLANE;               // hi + lo
constexpr float XLOG2E = 1.4426950408889634f, XLN2 = 0.6931471805599453f;

CSN_DEVINL f32x16 x4_mma(s16x8 ah, s16x8 al, s16x8 bh, s16x8 bl, f32x16 c) {
  c = mfma32<false>(al, bh, c);
  c = mfma32<false>(ah, bl, c);
  return mfma32<false>(ah, bh, c);
}
// One piece of a tile image straight into LDS (buffer_load ... lds: no staging registers): the wave writes 1 KB = 16 rows of 64
// bytes of one plane at lds_addr + 16 * lane; which bytes a lane fetches is its own source address (voff).  M0 carries the LDS
// address; it is saved and restored inside the statement (the compiler owns it).  Completion: the wave's vmcnt, then a barrier.
CSN_DEVINL void x4_dma(unsigned lds_addr, u32x4 rsrc, unsigned voff, unsigned soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff) : "memory");
}
template <int N>
CSN_DEVINL void x4_landed() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
CSN_DEVINL u32x4 x4_rsrc(const void* base, long long bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned nb = bytes > 0x7fffffffLL ? 0x7fffffffu : (bytes < 0 ? 0u : (unsigned)bytes);
  return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, nb, 0x00020000u};
}

// DROP: attention-probability dropout live; KEEP: the raw scores are stored for the backward
template <bool DROP, bool KEEP>
__global__ __launch_bounds__(256, 1) void csn_attn_fwd_x4_kernel(CsnAttnArgs p) {
  // [K stage 0 | K stage 1 | V stage 0 | V stage 1], each hi + lo planes of [256 rows][32 keys]: 128 KB — also the
  // [256][128 queries] fp32 block through which Qs comes in and Ctx goes out as 16-byte rows
  __shared__ __attribute__((aligned(16))) short tiles[4 * XIMG];
  float* xbuf = reinterpret_cast<float*>(tiles);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;

  // work-group -> (evaluation, head, block, query tile), XCD-aware as in attn_bf16x3.hip
  const int T_lay = p.T, Tp = p.Tp, ld = p.ld;
  const int QT = (T_lay + 127) / 128;
  const int Y = p.n_blocks * p.H;
  const int Lb = blockIdx.x, slot8 = Lb & 7, jj = Lb >> 3;
  const int qt = jj % QT, u = (jj / QT) * 8 + slot8;
  if (u >= Y * p.E) return;
  const int e = p.eval_ids ? p.eval_ids[u / Y] : u / Y;
  const int hd = (u % Y) % p.H, blk = (u % Y) / p.H;
  const bool short_blk = p.T_last > 0 && blk == p.n_blocks - 1;
  const int T = short_blk ? p.T_last : p.T;                     // queries = keys of this block
  if (qt * 128 >= T) return;
  const int qrow = qt * 128 + wave * 32 + l31;
  const bool q_ok = qrow < T;
  const long long qs = p.q_index ? p.q_index[e] : e, ks = p.kv_index ? p.kv_index[e] : e;
  const long long head_off = (long long)hd * XD * ld + (long long)blk * T_lay;
  const long long win = ((long long)(XD - 1) * ld + T_lay) * 4;
  const long long stat_off = ((long long)e * p.H + hd) * ((long long)p.n_blocks * T_lay) + (long long)blk * T_lay;

  // ---- Qs^T block [256][128 queries] -> LDS (16-byte rows) -> B fragments: lane (q, h) holds Qs[16 s + 8 h + j][q] ------------
  s16x8 Qh[XD / 16], Ql[XD / 16];
  {
    const csn_rsrc_t Rr = csn_make_rsrc(p.q + qs * p.q_shape_stride + head_off, win);
    const int cc = tid & 31, crow = tid >> 5;                   // 16-byte chunk column (4 queries), first row (rows + 8 t)
    const unsigned c_off = (qt * 128 + 4 * cc) < T ? (unsigned)(crow * ld + qt * 128 + 4 * cc) * 4u : CSN_OOB;
#pragma unroll
    for (int t8 = 0; t8 < 4; ++t8) {
      f32x4 ch[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) ch[t] = csn_bload4(Rr, c_off, (unsigned)((8 * (8 * t8 + t)) * ld) * 4u);
#pragma unroll
      for (int t = 0; t < 8; ++t) *reinterpret_cast<f32x4*>(&xbuf[(crow + 8 * (8 * t8 + t)) * 128 + 4 * cc]) = ch[t];
    }
    __syncthreads();
    const int col = 32 * wave + l31;
#pragma unroll
    for (int s = 0; s < XD / 16; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = xbuf[(16 * s + 8 * h + j) * 128 + col];
        Qh[s][j] = to16<false>(v);
        Ql[s][j] = to16<false>(v - from16<false>(Qh[s][j]));
      }
    __syncthreads();                                            // the staging block becomes the tile images
  }

  // ---- K / V tile planes: per row (d or channel) and block 16 tiles of [hi 32 keys | lo 32 keys] ----------------------------
  const int nkt = (T + XKT - 1) / XKT, nkt_all = nkt;
  const int kld = p.kv_ld;
  const long long kv_off = ks * p.kv_shape_stride + (long long)hd * XD * kld + (long long)blk * 1024;
  const long long kv_win = ((long long)(XD - 1) * kld + 1024) * 2;
  const u32x4 Kr = x4_rsrc(reinterpret_cast<const short*>(p.k) + kv_off, kv_win);
  const u32x4 Vr = x4_rsrc(reinterpret_cast<const short*>(p.v) + kv_off, kv_win);
  // staging by LDS-DMA: an image is 2 planes x 256 rows x 64 bytes = 32 pieces of 1 KB (16 rows of one plane); wave w moves
  // pieces 8 w .. 8 w + 7 of the K image and of the V image of a tile.  Lane -> row lane / 4 of the piece, 16-byte unit lane % 4.
  // K image: plain rows — the transposing read of a 32-lane half takes 4 consecutive 64-byte rows, every bank once.
  // V image: unit u of row r sits at u ^ ((r >> 2) & 3), applied on the SOURCE side (the DMA writes linearly): the 8-byte
  // reads of the second product (32 rows, one key chunk) then spread over 16 of the 32 eight-byte slots — a 2-way conflict
  // (an 8-byte swizzle would be conflict-free and is out of a 16-byte DMA's reach).
  typedef short __attribute__((address_space(3))) lds_short;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((lds_short*)tiles));
  const unsigned k_voff = (unsigned)((lane >> 2) * kld * 2 + (lane & 3) * 16);
  const unsigned v_voff = (unsigned)((lane >> 2) * kld * 2 + (((lane & 3) ^ ((lane >> 4) & 3))) * 16);
  // piece i (0..7) of this wave's share of an image; a tile beyond the block: every lane off — the piece is still issued, so
  // that the loop's wait count never changes
  auto dma_piece = [&](const u32x4& rs, unsigned voff, int img /* 0,1: K stages; 2,3: V stages */, int kt, int i) {
    const unsigned off = kt < nkt_all ? voff : CSN_OOB;
    const int pidx = 8 * wave + i, plane = pidx >> 4, rb = pidx & 15;
    x4_dma(lds0 + (unsigned)(img * XIMG + pidx * 512) * 2u, rs, off, (unsigned)(kt * 128 + plane * 64 + rb * 16 * kld * 2));
  };
  auto dma_tile = [&](const u32x4& rs, unsigned voff, int img, int kt) {
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_piece(rs, voff, img, kt, i);
  };

  // fragment addresses (lane constants).  K (A operand of S^T = K Qs^T): lane l holds K[key l & 31][16 s + 8 h + j] — 16-lane
  // group g covers keys 16 (g & 1) .. + 15 and rows 8 (g >> 1) .. + 7 in two passes of 4 rows; lane 4 q' + p' addresses row q',
  // keys 4 p' .. + 3
  const int grp = lane >> 4, gq = (lane >> 2) & 3, gp = lane & 3;
  const int k_base = (8 * (grp >> 1) + gq) * XKT + 16 * (grp & 1) + 4 * gp;
  // V (A operand of O^T += V^T P): lane l holds V[channel 32 t + (l & 31)][keys 16 s + 4 h .. + 3, 16 s + 8 + 4 h .. + 3]
  const int vg = (l31 >> 2) & 3;
  const int v_base = l31 * XKT + 4 * h;

  const bool drop = DROP;
  const unsigned thr16 = csn_drop_threshold16(p.dropout_p);
  const float keep_scale = DROP ? 1.f / (1.f - p.dropout_p) : 1.f;
  const unsigned salt = csn_block_salt((unsigned long long)(((long long)e * p.H + hd) * p.n_blocks + blk), p.seed);
  const int mp = T_lay > Tp ? T_lay : Tp;
  const long long sc_off = (((long long)e * p.H + hd) * p.n_blocks + blk) * ((long long)T_lay * Tp);
  const csn_rsrc_t Sr = csn_make_rsrc(KEEP ? p.scores + sc_off : nullptr, KEEP ? (long long)T_lay * Tp * 4 : 0);

  f32x16 O[XD / 32];
#pragma unroll
  for (int t = 0; t < XD / 32; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[t][r] = 0.f;
  float m_run = -INFINITY, m2_run = -INFINITY, l_run = 0.f;

  // S^T(kt) = K(kt) Qs^T from K stage st: reg r of lane (q, h) = key (r & 3) + 8 (r >> 2) + 4 h of query q
  // (fragment reads run XPD steps ahead of the matrix instructions that use them: with one wave per SIMD nothing else hides
  //  the ~130 cycles of an LDS read, and a step is only 96 matrix-pipe cycles)
  constexpr int XPD = 2;
  // kt_dma >= 0: the 16 DMA pieces of the next tiles — K(kt_dma + 2) into K stage k_img, V(kt_dma + 1) into V stage v_img — are
  // issued one per k step BETWEEN the matrix instructions (16 in a burst at the top of the iteration stalled the wave's one
  // instruction stream for 2.5 k cycles: stamps), early enough to land under the second product
  auto phase1 = [&](int st, int kt_dma, int k_img, int v_img) {
    f32x16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
    const short* kh = tiles + st * XIMG + k_base;
    s16x8 fh[XPD], fl[XPD];
    auto rd = [&](int s, s16x8& a_h, s16x8& a_l) {
      const short* a = kh + 16 * s * XKT;
      a_h = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a)),
                  __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + 4 * XKT)));
      a_l = join8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + XPLANE)),
                  __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + XPLANE + 4 * XKT)));
    };
#pragma unroll
    for (int s = 0; s < XPD; ++s) rd(s, fh[s], fl[s]);
#pragma unroll
    for (int s = 0; s < XD / 16; ++s) {
      const int r = s % XPD;
      S = x4_mma(fh[r], fl[r], Qh[s], Ql[s], S);
      if (s + XPD < XD / 16) rd(s + XPD, fh[r], fl[r]);
      if (kt_dma >= 0) {
        if (s < 8) dma_piece(Kr, k_voff, k_img, kt_dma + 2, s);
        else dma_piece(Vr, v_voff, v_img, kt_dma + 1, s - 8);
      }
    }
    return S;
  };
  // O^T += V^T(kt) P: 8 channel tiles x 2 key steps
  auto phase2 = [&](int st, const s16x8* Ph, const s16x8* Pl) {
    const short* vh = tiles + (2 + st) * XIMG + v_base;
    s16x8 fh[XPD], fl[XPD];
    auto rd = [&](int i, s16x8& a_h, s16x8& a_l) {
      const int t = i >> 1, s = i & 1;
      const short* a = vh + 32 * t * XKT;
      const int c0 = 8 * ((2 * s) ^ vg), c1 = 8 * ((2 * s + 1) ^ vg);
      a_h = join8(*reinterpret_cast<const s16x4*>(a + c0), *reinterpret_cast<const s16x4*>(a + c1));
      a_l = join8(*reinterpret_cast<const s16x4*>(a + XPLANE + c0), *reinterpret_cast<const s16x4*>(a + XPLANE + c1));
    };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < XPD; ++i) rd(i, fh[i], fl[i]);
    __builtin_amdgcn_sched_group_barrier(0x100, 4 * XPD, 0);
#pragma unroll
    for (int i = 0; i < XD / 16; ++i) {
      const int r = i % XPD, t = i >> 1, s = i & 1;
      O[t] = x4_mma(fh[r], fl[r], Ph[s], Pl[s], O[t]);
      if (i + XPD < XD / 16) rd(i + XPD, fh[r], fl[r]);
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: K(0), K(1), V(0) into LDS, S(0) ----------------------------------------------------------------------------
  dma_tile(Kr, k_voff, 0, 0);
  dma_tile(Vr, v_voff, 2, 0);
  dma_tile(Kr, k_voff, 1, 1);
  x4_landed<0>();
  __syncthreads();
  f32x16 S = phase1(0, -1, 0, 0);

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1, nxt = cur ^ 1;
#ifdef CSN_X4_STAMPS
    const bool dbg_on = blockIdx.x >= 2048 && blockIdx.x < 2048 + 256 && kt >= 4 && kt < 12;
    unsigned long long stamps[8];
#endif
    XSTAMP(0);
    // (tiles of the next iteration: K(kt + 2) replaces K(kt), read in the previous iteration; V(kt + 1) replaces V(kt - 1) —
    //  nobody reads either image during this iteration; their DMA pieces are issued inside the first matrix phase below)
    XSTAMP(1);

    // keys beyond the block's end exist only in the last tile: scores -inf there (the planes hold zeros)
    if (kt == nkt - 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) S[r] = (kt * XKT + (r & 3) + 8 * (r >> 2) + 4 * h) < T ? S[r] : -INFINITY;
    }
    // lazy re-basing of the running maximum (the two lane halves of a query share m_run; the cross-lane maximum only in the
    // rare branch)
    {
      float mx = S[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[r]);
      if (__any(mx > m_run + p.rescale_threshold)) {
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m2_new = m_new * XLOG2E;
        const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m2_run - m2_new);
#pragma unroll
        for (int t = 0; t < XD / 32; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) O[t][r] *= alpha;
        l_run *= alpha;
        m_run = m_new;
        m2_run = m2_new;
      }
    }
    XSTAMP(2);
    // ---- S(kt + 1) on the matrix pipe beside the pointwise work of tile kt on the vector pipe --------------------------------
    // (the last iteration contracts a stale K image: finite numbers, never used)
    f32x16 Sn = phase1(nxt, kt, cur, 2 + nxt);
    if (KEEP) {                                                 // the raw scores, [query][key] rows: 4 keys = 16 bytes per group
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const bool ok = q_ok && (kt * XKT + 8 * g + 4 * h) < T;
        const unsigned off = ok ? (unsigned)(qrow * Tp + kt * XKT + 8 * g + 4 * h) * 4u : CSN_OOB;
        csn_bstore4_stream(f32x4{S[4 * g], S[4 * g + 1], S[4 * g + 2], S[4 * g + 3]}, Sr, off);
      }
    }
    float pr[16];
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pr[r] = __builtin_amdgcn_exp2f(fmaf(S[r], XLOG2E, -m2_run));       // masked keys: exp2(-inf) = 0
      ps += pr[r];
    }
    l_run += ps;                                                 // the denominator sees every key, dropped or not
    if (drop) {
      // one hash per key pair (csn_common.h): keys 2 w, 2 w + 1 of query q have pair index w * mp + q
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2) {
          const unsigned w = (unsigned)(kt * (XKT / 2) + 4 * g + 2 * h + w2);
          const unsigned hsh = csn_pair_hash(w * (unsigned)mp + (unsigned)qrow, salt);
          const int r0 = 4 * g + 2 * w2;
          pr[r0] = (hsh & 0xffffu) >= thr16 ? pr[r0] * keep_scale : 0.f;
          pr[r0 + 1] = (hsh >> 16) >= thr16 ? pr[r0 + 1] * keep_scale : 0.f;
        }
    }
    s16x8 Ph[2], Pl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        Ph[s][j] = to16<false>(pr[8 * s + j]);
        Pl[s][j] = to16<false>(pr[8 * s + j] - from16<false>(Ph[s][j]));
      }
    XSTAMP(3);
    // ---- O += V(kt) P(kt) ---------------------------------------------------------------------------------------------------
    phase2(cur, Ph, Pl);
    XSTAMP(4);
    // ---- the next tiles have landed: behind their pieces only this iteration's four score stores were issued ----------------
    x4_landed<KEEP ? 4 : 0>();
    XSTAMP(5);
    S = Sn;
    __syncthreads();
    XSTAMP(6);
#ifdef CSN_X4_STAMPS
    if (dbg_on && lane == 0)
      for (int i = 0; i < 7; ++i) csn_x4_dbg[(((blockIdx.x - 2048) * 4 + wave) * 8 + (kt - 4)) * 8 + i] = stamps[i];
#endif
  }

  // ---- epilogue: lse, Ctx^T through the [256][128] block as 16-byte rows ------------------------------------------------------
  float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  if (q_ok && h == 0 && p.lse) p.lse[stat_off + qrow] = m2_run * XLN2 + logf(l_tot);
  {
    const int col = 32 * wave + l31;
#pragma unroll
    for (int t = 0; t < XD / 32; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) xbuf[(32 * t + csn_acc_row(r, h)) * 128 + col] = O[t][r] * inv;
    __syncthreads();
    const long long os = p.out_index ? p.out_index[e] : e;
    const csn_rsrc_t Or = csn_make_rsrc(p.out + os * p.out_eval_stride + head_off, win);
    const int cc = tid & 31, crow = tid >> 5;
    const unsigned c_off = (qt * 128 + 4 * cc) < T ? (unsigned)(crow * ld + qt * 128 + 4 * cc) * 4u : CSN_OOB;
#pragma unroll
    for (int t8 = 0; t8 < 4; ++t8) {
      f32x4 ch[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) ch[t] = *reinterpret_cast<const f32x4*>(&xbuf[(crow + 8 * (8 * t8 + t)) * 128 + 4 * cc]);
#pragma unroll
      for (int t = 0; t < 8; ++t) csn_bstore4(ch[t], Or, c_off, (unsigned)((8 * (8 * t8 + t)) * ld) * 4u);
    }
  }
}

}  // namespace

// MEASURED AND NOT TAKEN (round 4, profiles/r4_x4_forward.txt): 3.19 ms against 2.55 ms for the eight-wave kernel at 128
// evaluations of config-3 geometry.  The stamps say why: with ONE wave per SIMD every wave-level memory instruction stalls the
// SIMD's only instruction stream for ~150 cycles (16 DMA pieces per tile: 2.5-2.9 k cycles, as much as the tile's 3.07 k cycles
// of matrix instructions) and every LDS round trip is exposed (the second product runs at 3.4 k cycles for 1.5 k of matrix
// work) — exactly what a SIMD partner hides in the eight-wave kernel.  Kept behind the switch as the measured form of the
// "32 queries per wave on 32x32x16" experiment; tests/test_gpu_attn_x4.py holds it to the default kernel's results.
int csn_dev_attn_x4 = 0;      // development switch (csn_dev_set CSN_DEV_ATTN_X4): 1 = d = 256 bf16x3 forwards on this kernel

bool csn_attn_fwd_x4_takes(const CsnAttnArgs& a, int d, int mode) {
  return csn_dev_attn_x4 != 0 && mode == 1 && d == 256 && a.kv_planes && a.Tq == 0 && !a.tq_arr && !a.t_arr && !a.sc_layout &&
         !a.r_fmt && !a.out_fmt && !a.accumulate && !a.grp_off && a.T <= 512 && a.T_last <= a.T;
}

int csn_launch_attn_fwd_x4(const CsnAttnArgs& a, hipStream_t st) {
  if (a.E <= 0 || a.n_blocks <= 0) return 0;
  if ((a.ld & 3) || (a.Tp & 3) || (a.T & 3) || (a.T_last & 3) || (a.kv_ld & 7) || (a.kv_shape_stride & 7) || (a.q_shape_stride & 3)) return -2;
  const long long units = (long long)a.n_blocks * a.H * a.E;
  dim3 grid((unsigned)(((units + 7) / 8) * 8 * ((a.T + 127) / 128)));
  const bool drop = a.dropout_p > 0.f, keep = a.scores != nullptr;
  if (drop && keep) hipLaunchKernelGGL((csn_attn_fwd_x4_kernel<true, true>), grid, dim3(256), 0, st, a);
  else if (drop) hipLaunchKernelGGL((csn_attn_fwd_x4_kernel<true, false>), grid, dim3(256), 0, st, a);
  else if (keep) hipLaunchKernelGGL((csn_attn_fwd_x4_kernel<false, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((csn_attn_fwd_x4_kernel<false, false>), grid, dim3(256), 0, st, a);
  return (int)hipGetLastError();
}
