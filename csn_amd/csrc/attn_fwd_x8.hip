// Fused block attention FORWARD at d_k = d_v = 256 in the bf16x3 mode on v_mfma_f32_32x32x16_bf16 with TWO waves per SIMD: eight
// waves = four query groups of 32 queries x two halves of the 256 channels (MID-FC/csa_models.py:138-144, 147).
//
// attn_fwd_x4.hip showed what 32 queries per wave buy (every LDS fragment feeds six matrix instructions instead of three, half the
// LDS bytes per FLOP) and what one wave per SIMD costs (nothing hides a wave-level memory instruction or an LDS round trip).  This
// form keeps the first and gives the SIMD its partner back: the two waves of a query group split the CHANNELS —
//   * S^T = K Qs^T: each wave contracts its 128 channels (24 matrix instructions per 32-key tile, its half of the K image), the
//     halves meet through LDS (4 KB per wave, the same lanes and registers on both sides) and both waves hold the whole tile;
//   * softmax, dropout, hi / lo split: done by both waves of the pair (the same bits) — vector work for LDS traffic;
//   * O^T += V^T P: each wave owns its 128 channels of the output (24 matrix instructions, its half of the V image).
// Registers per wave: Qs^T 64 (hi / lo B fragments of 128 channels), O 64, two score tiles 32 — inside 256.  Two barriers per
// tile (the exchange buffer is single: LDS is full).  K / V staging, images, key order of P, masks, lazy re-basing, lse: as in
// attn_fwd_x4.hip, whose helpers this file shares.
#include "csn_common.h"
#include "csn_kernels.h"
#define XSTAMP(i)

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
__global__ __launch_bounds__(512, 2) void csn_attn_fwd_x8_kernel(CsnAttnArgs p) {
  // [K stage 0 | K stage 1 | V stage 0 | V stage 1], each hi + lo planes of [256 rows][32 keys]: 128 KB — also the
  // [256][128 queries] fp32 block through which Qs comes in and Ctx goes out as 16-byte rows — and 8 x 4 KB through which the
  // two waves of a query group exchange their halves of a score tile: 160 KB, the whole LDS of the CU
  __shared__ __attribute__((aligned(16))) short tiles[4 * XIMG + 8 * 2048];
  float* xbuf = reinterpret_cast<float*>(tiles);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int qg = wave >> 1, dh = wave & 1;                      // query group (32 queries), half of the 256 channels
  float* xch_own = reinterpret_cast<float*>(tiles + 4 * XIMG) + wave * 1024 + 4 * lane;
  const float* xch_partner = reinterpret_cast<const float*>(tiles + 4 * XIMG) + (wave ^ 1) * 1024 + 4 * lane;

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
  const int qrow = qt * 128 + qg * 32 + l31;
  const bool q_ok = qrow < T;
  const long long qs = p.q_index ? p.q_index[e] : e, ks = p.kv_index ? p.kv_index[e] : e;
  const long long head_off = (long long)hd * XD * ld + (long long)blk * T_lay;
  const long long win = ((long long)(XD - 1) * ld + T_lay) * 4;
  const long long stat_off = ((long long)e * p.H + hd) * ((long long)p.n_blocks * T_lay) + (long long)blk * T_lay;

  // ---- Qs^T block [256][128 queries] -> LDS (16-byte rows) -> B fragments: lane (q, h) holds Qs[128 dh + 16 s + 8 h + j][q] ---
  constexpr int XS = XD / 32;                                   // k steps of this wave's half of the channels
  s16x8 Qh[XS], Ql[XS];
  {
    const csn_rsrc_t Rr = csn_make_rsrc(p.q + qs * p.q_shape_stride + head_off, win);
    const int cc = tid & 31, crow = tid >> 5;                   // 16-byte chunk column (4 queries), first row (rows + 16 t)
    const unsigned c_off = (qt * 128 + 4 * cc) < T ? (unsigned)(crow * ld + qt * 128 + 4 * cc) * 4u : CSN_OOB;
#pragma unroll
    for (int t8 = 0; t8 < 2; ++t8) {
      f32x4 ch[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) ch[t] = csn_bload4(Rr, c_off, (unsigned)((16 * (8 * t8 + t)) * ld) * 4u);
#pragma unroll
      for (int t = 0; t < 8; ++t) *reinterpret_cast<f32x4*>(&xbuf[(crow + 16 * (8 * t8 + t)) * 128 + 4 * cc]) = ch[t];
    }
    __syncthreads();
    const int col = 32 * qg + l31;
#pragma unroll
    for (int s = 0; s < XS; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = xbuf[(128 * dh + 16 * s + 8 * h + j) * 128 + col];
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
  // pieces 4 w .. 4 w + 3 of the K image and of the V image of a tile.  Lane -> row lane / 4 of the piece, 16-byte unit lane % 4.
  // K image: plain rows — the transposing read of a 32-lane half takes 4 consecutive 64-byte rows, every bank once.
  // V image: unit u of row r sits at u ^ ((r >> 2) & 3), applied on the SOURCE side (the DMA writes linearly): the 8-byte
  // reads of the second product (32 rows, one key chunk) then spread over 16 of the 32 eight-byte slots — a 2-way conflict
  // (an 8-byte swizzle would be conflict-free and is out of a 16-byte DMA's reach).
  typedef short __attribute__((address_space(3))) lds_short;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((lds_short*)tiles));
  const unsigned k_voff = (unsigned)((lane >> 2) * kld * 2 + (lane & 3) * 16);
  const unsigned v_voff = (unsigned)((lane >> 2) * kld * 2 + (((lane & 3) ^ ((lane >> 4) & 3))) * 16);
  // piece i (0..3) of this wave's share of an image; a tile beyond the block: every lane off — the piece is still issued, so
  // that the loop's wait count never changes
  auto dma_piece = [&](const u32x4& rs, unsigned voff, int img /* 0,1: K stages; 2,3: V stages */, int kt, int i) {
    const unsigned off = kt < nkt_all ? voff : CSN_OOB;
    const int pidx = 4 * wave + i, plane = pidx >> 4, rb = pidx & 15;
    x4_dma(lds0 + (unsigned)(img * XIMG + pidx * 512) * 2u, rs, off, (unsigned)(kt * 128 + plane * 64 + rb * 16 * kld * 2));
  };
  // inside the loop the halves of the work-group split the two images: waves 0..3 move the K image (8 pieces each), waves 4..7
  // — which run one barrier segment behind (below) — the V image
  const bool late = wave >= 4;
  auto dma_loop_piece = [&](int kt, int k_img, int v_img, int i) {
    const int pidx = 8 * (wave & 3) + i, plane = pidx >> 4, rb = pidx & 15;
    const int kt_t = late ? kt + 1 : kt + 2, img = late ? v_img : k_img;
    const unsigned off = kt_t < nkt_all ? (late ? v_voff : k_voff) : CSN_OOB;
    x4_dma(lds0 + (unsigned)(img * XIMG + pidx * 512) * 2u, late ? Vr : Kr, off, (unsigned)(kt_t * 128 + plane * 64 + rb * 16 * kld * 2));
  };
  auto dma_tile = [&](const u32x4& rs, unsigned voff, int img, int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(rs, voff, img, kt, i);
  };

  // fragment addresses (lane constants).  K (A operand of S^T = K Qs^T): lane l holds K[key l & 31][16 s + 8 h + j] — 16-lane
  // group g covers keys 16 (g & 1) .. + 15 and rows 8 (g >> 1) .. + 7 in two passes of 4 rows; lane 4 q' + p' addresses row q',
  // keys 4 p' .. + 3
  const int grp = lane >> 4, gq = (lane >> 2) & 3, gp = lane & 3;
  const int k_base = (128 * dh + 8 * (grp >> 1) + gq) * XKT + 16 * (grp & 1) + 4 * gp;
  // V (A operand of O^T += V^T P): lane l holds V[channel 32 t + (l & 31)][keys 16 s + 4 h .. + 3, 16 s + 8 + 4 h .. + 3]
  const int vg = (l31 >> 2) & 3;
  const int v_base = (128 * dh + l31) * XKT + 4 * h;

  const bool drop = DROP;
  const unsigned thr16 = csn_drop_threshold16(p.dropout_p);
  const float keep_scale = DROP ? 1.f / (1.f - p.dropout_p) : 1.f;
  const unsigned salt = csn_block_salt((unsigned long long)(((long long)e * p.H + hd) * p.n_blocks + blk), p.seed);
  const int mp = T_lay > Tp ? T_lay : Tp;
  const long long sc_off = (((long long)e * p.H + hd) * p.n_blocks + blk) * ((long long)T_lay * Tp);
  const csn_rsrc_t Sr = csn_make_rsrc(KEEP ? p.scores + sc_off : nullptr, KEEP ? (long long)T_lay * Tp * 4 : 0);

  f32x16 O[XD / 64];                                            // this wave's 128 channels of its 32 queries
#pragma unroll
  for (int t = 0; t < XD / 64; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[t][r] = 0.f;
  float m_run = -INFINITY, m2_run = -INFINITY, l_run = 0.f;

  // this wave's HALF of S^T(kt) = K(kt) Qs^T — the sum over its 128 channels — from K stage st: reg r of lane (q, h) = key
  // (r & 3) + 8 (r >> 2) + 4 h of query q; the partner wave holds the other half in the same lanes and registers
  // (fragment reads run XPD steps ahead of the matrix instructions that use them: with one wave per SIMD nothing else hides
  //  the ~130 cycles of an LDS read, and a step is only 96 matrix-pipe cycles)
  constexpr int XPD = 2;
  // kt_dma >= 0: the wave's 8 DMA pieces of the next tiles — K(kt_dma + 2) into K stage k_img, V(kt_dma + 1) into V stage v_img —
  // are issued one per k step between the matrix instructions, early enough to land under the second product
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
    for (int s = 0; s < XS; ++s) {
      const int r = s % XPD;
      S = x4_mma(fh[r], fl[r], Qh[s], Ql[s], S);
      if (s + XPD < XS) rd(s + XPD, fh[r], fl[r]);
      if (kt_dma >= 0 && !(p.dev_ablate & 1)) dma_loop_piece(kt_dma, k_img, v_img, s);
    }
    return S;
  };
  // O^T += V^T(kt) P: this wave's 4 channel tiles x 2 key steps
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
    for (int i = 0; i < XD / 32; ++i) {
      const int r = i % XPD, t = i >> 1, s = i & 1;
      O[t] = x4_mma(fh[r], fl[r], Ph[s], Pl[s], O[t]);
      if (i + XPD < XD / 32) rd(i + XPD, fh[r], fl[r]);
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
  // the halves of a score tile meet through LDS: lane l of a wave and lane l of its partner hold the same (query, keys)
  auto put_half = [&](const f32x16& part) {
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(xch_own + 256 * g) = f32x4{part[4 * g], part[4 * g + 1], part[4 * g + 2], part[4 * g + 3]};
  };
  auto add_halves = [&](const f32x16& own) {
    f32x16 full;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 o = *reinterpret_cast<const f32x4*>(xch_partner + 256 * g);
#pragma unroll
      for (int i = 0; i < 4; ++i) full[4 * g + i] = own[4 * g + i] + o[i];        // (a + b = b + a: both waves get the same bits)
    }
    return full;
  };
  f32x16 Sn = phase1(0, -1, 0, 0);
  put_half(Sn);
  __syncthreads();
  f32x16 S = add_halves(Sn);
  // Staggered halves: waves 4..7 — the SIMD partners of waves 0..3, and other query groups — run one barrier segment behind, so
  // that a SIMD has one wave in its first segment (S product + pointwise work) beside one in its second (O product), instead of
  // two waves issuing the same kind of work in lock step.  K(kt + 2) is requested by the early half in its first segment (the
  // stage's last readers: both halves' first segments of the previous tile, over by then); V(kt + 1) by the late half in ITS
  // first segment (the stage's last reader is the late half's own second segment of the previous tile).
  if (late) __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1, nxt = cur ^ 1;
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
        for (int t = 0; t < XD / 64; ++t)
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
    Sn = phase1(nxt, kt, cur, 2 + nxt);
    if (KEEP && !(p.dev_ablate & 4)) {                          // the raw scores, [query][key] rows: 4 keys = 16 bytes per group;
#pragma unroll                                                  // each wave of the pair stores two of the four groups
      for (int i = 0; i < 2; ++i) {
        const int g = 2 * dh + i;
        const bool ok = q_ok && (kt * XKT + 8 * g + 4 * h) < T;
        const unsigned off = ok ? (unsigned)(qrow * Tp + kt * XKT + 8 * g + 4 * h) * 4u : CSN_OOB;
        const f32x4 lo4 = f32x4{S[4 * i], S[4 * i + 1], S[4 * i + 2], S[4 * i + 3]};
        const f32x4 hi4 = f32x4{S[8 + 4 * i], S[8 + 4 * i + 1], S[8 + 4 * i + 2], S[8 + 4 * i + 3]};
        csn_bstore4_stream(dh ? hi4 : lo4, Sr, off);
      }
    }
    float pr[16];
    float ps = 0.f;
    if (p.dev_ablate & 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) pr[r] = S[r];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pr[r] = __builtin_amdgcn_exp2f(fmaf(S[r], XLOG2E, -m2_run));       // masked keys: exp2(-inf) = 0
        ps += pr[r];
      }
    }
    l_run += ps;                                                 // the denominator sees every key, dropped or not
    if (drop && !(p.dev_ablate & 2)) {
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
    if (p.dev_ablate & 8) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) { Ph[s][j] = (short)__builtin_bit_cast(int, pr[8 * s + j]); Pl[s][j] = Ph[s][j]; }
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          Ph[s][j] = to16<false>(pr[8 * s + j]);
          Pl[s][j] = to16<false>(pr[8 * s + j] - from16<false>(Ph[s][j]));
        }
    }
    XSTAMP(3);
    __syncthreads();                                            // every wave has read its partner's half of S(kt)
    put_half(Sn);
    // ---- O += V(kt) P(kt) ---------------------------------------------------------------------------------------------------
    phase2(cur, Ph, Pl);
    XSTAMP(4);
    // ---- the next tiles have landed: behind their pieces only this iteration's two score stores were issued -----------------
    x4_landed<KEEP ? 2 : 0>();
    XSTAMP(5);
    __syncthreads();
    S = add_halves(Sn);
    XSTAMP(6);
  }

  if (!late) __syncthreads();
  // ---- epilogue: lse, Ctx^T through the [256][128] block as 16-byte rows ------------------------------------------------------
  float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  if (q_ok && h == 0 && dh == 0 && p.lse) p.lse[stat_off + qrow] = m2_run * XLN2 + logf(l_tot);
  {
    const int col = 32 * qg + l31;
#pragma unroll
    for (int t = 0; t < XD / 64; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) xbuf[(128 * dh + 32 * t + csn_acc_row(r, h)) * 128 + col] = O[t][r] * inv;
    __syncthreads();
    const long long os = p.out_index ? p.out_index[e] : e;
    const csn_rsrc_t Or = csn_make_rsrc(p.out + os * p.out_eval_stride + head_off, win);
    const int cc = tid & 31, crow = tid >> 5;
    const unsigned c_off = (qt * 128 + 4 * cc) < T ? (unsigned)(crow * ld + qt * 128 + 4 * cc) * 4u : CSN_OOB;
#pragma unroll
    for (int t8 = 0; t8 < 2; ++t8) {
      f32x4 ch[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) ch[t] = *reinterpret_cast<const f32x4*>(&xbuf[(crow + 16 * (8 * t8 + t)) * 128 + 4 * cc]);
#pragma unroll
      for (int t = 0; t < 8; ++t) csn_bstore4(ch[t], Or, c_off, (unsigned)((16 * (8 * t8 + t)) * ld) * 4u);
    }
  }
}

}  // namespace

int csn_launch_attn_fwd_x8(const CsnAttnArgs& a, hipStream_t st) {
  if (a.E <= 0 || a.n_blocks <= 0) return 0;
  if ((a.ld & 3) || (a.Tp & 3) || (a.T & 3) || (a.T_last & 3) || (a.kv_ld & 7) || (a.kv_shape_stride & 7) || (a.q_shape_stride & 3)) return -2;
  const long long units = (long long)a.n_blocks * a.H * a.E;
  dim3 grid((unsigned)(((units + 7) / 8) * 8 * ((a.T + 127) / 128)));
  const bool drop = a.dropout_p > 0.f, keep = a.scores != nullptr;
  CsnAttnArgs b = a;
  b.dev_ablate = csn_dev_attn_x4 >> 4;                          // CSN_DEV_ATTN_X4 = 2 | bits << 4: 1 no staging requests in the loop,
  if (b.dev_ablate) {                                           // 2 no exp / dropout, 4 no score stores, 8 no hi / lo split
    hipLaunchKernelGGL((csn_attn_fwd_x8_kernel<true, true>), grid, dim3(512), 0, st, b);
    return (int)hipGetLastError();
  }
  if (drop && keep) hipLaunchKernelGGL((csn_attn_fwd_x8_kernel<true, true>), grid, dim3(512), 0, st, a);
  else if (drop) hipLaunchKernelGGL((csn_attn_fwd_x8_kernel<true, false>), grid, dim3(512), 0, st, a);
  else if (keep) hipLaunchKernelGGL((csn_attn_fwd_x8_kernel<false, true>), grid, dim3(512), 0, st, a);
  else hipLaunchKernelGGL((csn_attn_fwd_x8_kernel<false, false>), grid, dim3(512), 0, st, a);
  return (int)hipGetLastError();
}
