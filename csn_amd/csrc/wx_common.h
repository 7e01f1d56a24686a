// Helpers shared by the weight-stationary streaming kernels (wx_stream.hip, wx_lnb.hip): the matrix step, requests hidden from the
// compiler and waited for by hand, LDS-DMA, the store with its wait state, scalar loads, descriptors, the persistent grid.
#pragma once
#include "csn_common.h"
#include <hip/hip_runtime.h>

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

// The chunk requests are hidden from the compiler (inline asm) and waited for by hand.  hipcc's wait-count model for gfx950
// does not count STORES in vmcnt, the hardware does: with compiler-visible loads every commit waited "all but 4..7" —
// which, with this chunk's 4 stores behind them, drained the request made at the top of the same iteration as well (one chunk
// in flight instead of two: 4.1 TB/s).  An asm load's destination is untouched by the compiler until the wait statement that
// names it "+v"; the counts below are the stores and requests issued after the request being waited for, all unconditional.
CSN_DEVINL void wx_request(f32x4& dst, u32x4 rsrc, unsigned voff, unsigned soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(soff)) : "memory");
}
template <int N>
CSN_DEVINL void wx_arrived(f32x4* R) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]) : "n"(N) : "memory");
}
// a residual chunk goes from memory straight to the wave's LDS block (no registers): lane l's 16 bytes land at m0 + 16 l
CSN_DEVINL void wx_dma(unsigned lds_addr, u32x4 rsrc, unsigned voff, unsigned soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(lds_addr)), "s"(__builtin_amdgcn_readfirstlane(soff)) : "memory");
}
// 16-byte store whose data registers may be rewritten by the very next instruction.  Measured on gfx950: after
//   buffer_store_dwordx4 v[0:3], v10, s[44:47], s64 offen ;  v_add_f32 v0, v0, v1
// lanes 12..15 (mod 16) of the second wave of a SIMD stored the SUM in the first dword now and then.  hipcc separates the two
// with s_nop when the store's soffset is an immediate and not when it is a register; the hardware needs it in both cases.
CSN_DEVINL void wx_store4(f32x4 v, u32x4 rsrc, unsigned voff, unsigned soff) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" :: "v"(v), "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(soff)) : "memory");
}
// a wave-uniform int through the scalar cache (left to the compiler this became a vector load and a vmcnt(0) — which drains
// every chunk request in flight)
CSN_DEVINL int wx_sload(const int* ptr) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const unsigned long long u = ((unsigned long long)hi << 32) | lo;
  int v;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(u) : "memory");
  return v;
}
CSN_DEVINL u32x4 wx_rsrc(const void* base, long long bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned nb = bytes > 0x7fffffffLL ? 0x7fffffffu : (bytes < 0 ? 0u : (unsigned)bytes);
  // (wave-uniform by construction; said so to the compiler, which must keep the words in scalar registers for the asm operands)
  return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)),
               (unsigned)__builtin_amdgcn_readfirstlane((int)nb), 0x00020000u};
}

inline int wx_grid() {
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
