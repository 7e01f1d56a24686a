// Shared device-side helpers for the CSN cross-shape-attention kernels (gfx950 / CDNA4 only).
//
// Conventions used by every kernel in this directory
//   * All activations are CHANNEL-MAJOR: a tensor of per-point features is stored [channel][point]
//     exactly like the reference's (B, C, N, 1) input (MID-FC/csa_models.py:88-94 gathers along N).
//     Lanes always run along the point index, so every global access is a 128-byte row segment.
//   * Matrix products use the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32
//     (D = A(32x2) * B(2x32) + C, fp32 in, fp32 accumulate, bit-identical to an fmaf chain).
//       A operand : lane l holds A[i = l & 31][k = l >> 5]
//       B operand : lane l holds B[k = l >> 5][j = l & 31]
//       C/D       : reg r of lane l is C[i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][j = l & 31]
//   * The contraction index inside an 8-wide k group is visited in the order k = 4*h + t
//     (h = lane >> 5, t = 0..3) whenever an operand is fetched from LDS with one 16-byte read per
//     lane; both operands use the same order, so the sum is simply re-associated.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CSN_DEVINL __device__ __forceinline__

CSN_DEVINL f32x16 csn_mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// row index inside a 32x32 accumulator tile held by (reg r, lane half h)
CSN_DEVINL int csn_acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- buffer (SRD) addressing ---------------------------------------------------------------------
// Every global access of the hot kernels goes through a wave-uniform buffer descriptor plus a 32-bit
// per-lane byte offset: out-of-window lanes are dropped by the hardware range check (loads return 0,
// stores vanish), so ragged tiles need no branches and no 64-bit per-lane address arithmetic.
// A lane is switched off by giving it CSN_OOB as its offset.  Windows are kept below 2 GiB.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t csn_rsrc_t;
#define CSN_OOB 0x80000000u

CSN_DEVINL csn_rsrc_t csn_make_rsrc(const void* base, long long bytes) {
  const unsigned nb = bytes > 0x7fffffffLL ? 0x7fffffffu : (bytes < 0 ? 0u : (unsigned)bytes);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, nb, 0x00020000);
}
CSN_DEVINL float csn_bload(csn_rsrc_t r, unsigned voff, unsigned soff = 0) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
CSN_DEVINL f32x4 csn_bload4(csn_rsrc_t r, unsigned voff, unsigned soff = 0) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
CSN_DEVINL void csn_bstore(float v, csn_rsrc_t r, unsigned voff, unsigned soff = 0) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
// Store-data hazard (DESIGN, "platform findings"): on gfx950 a 12 / 16-byte buffer store needs TWO wait states before a vector
// instruction rewrites its data registers.  hipcc pads them when the store's soffset is an immediate and not when it is a
// register (the rule it inherits says the hazard exists only in the first case; measured on MI355X: ~600 of 400 000 elements
// of a LayerNorm epilogue stored the NEXT value).  Every 16-byte store of this library whose soffset is not a compile-time
// constant is therefore followed by a guard: an `s_nop 1` that names the data registers as an input, so no instruction that
// rewrites them can be scheduled between the store and the end of the two wait states.  tests/test_cpu_store_hazard.py scans the
// gfx950 assembly of every shipped source for a site the guard does not cover.
CSN_DEVINL void csn_store_guard(f32x4 v) { asm volatile("s_nop 1" :: "v"(v)); }
CSN_DEVINL void csn_bstore4(f32x4 v, csn_rsrc_t r, unsigned voff, unsigned soff = 0) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
  if (!__builtin_constant_p(soff)) csn_store_guard(v);
}
// streaming forms for bytes that are written once and read once much later (the saved scores, the P / dS planes).
// -DCSN_NT=1 gives them the nt cache policy (aux bit 1) so that they do not push the K / V tiles out of the XCD's L2:
// measured over the whole step it changes nothing in time (28.93 vs 28.94 ms, same box) and WRITES 9 % MORE to HBM
// (the two 16-byte stores a lane makes into one 32-byte sector are no longer merged in L2) — off by default.
#ifndef CSN_NT
#define CSN_NT 0
#endif
CSN_DEVINL f32x4 csn_bload4_stream(csn_rsrc_t r, unsigned voff, unsigned soff = 0) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, CSN_NT ? 2 : 0));
}
CSN_DEVINL void csn_bstore4_stream(f32x4 v, csn_rsrc_t r, unsigned voff, unsigned soff = 0) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, CSN_NT ? 2 : 0);
  if (!__builtin_constant_p(soff)) csn_store_guard(v);
}
// 8-byte (4 x bf16) and 2-byte (1 x bf16) accesses for split-bf16 planes
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
CSN_DEVINL u32x2 csn_bload2(csn_rsrc_t r, unsigned voff, unsigned soff = 0) {
  return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
}
CSN_DEVINL void csn_bstore_bf16(__bf16 v, csn_rsrc_t r, unsigned voff, unsigned soff = 0) {
  __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, v), r, voff, soff, 0);
}
CSN_DEVINL void csn_bstore16(short v, csn_rsrc_t r, unsigned voff, unsigned soff = 0) {
  __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, r, voff, soff, 0);
}

// Guarded 16-byte load of 4 consecutive floats row[c..c+3]; elements at index >= clim read as 0.
// row + c must be 16-byte aligned when c + 3 < clim (hosts check ld % 4 == 0 and offsets % 4 == 0).
CSN_DEVINL f32x4 csn_ldg4(const float* __restrict__ row, int c, int clim, bool row_ok) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (row_ok) {
    if (c + 3 < clim) {
      v = *reinterpret_cast<const f32x4*>(row + c);
    } else {
      if (c < clim) v.x = row[c];
      if (c + 1 < clim) v.y = row[c + 1];
      if (c + 2 < clim) v.z = row[c + 2];
    }
  }
  return v;
}

// Three-level batched operand: element offset = s0*z0 + s1*z1 + s2*(idx2 ? idx2[z2] : z2)
// `planes` != 0 (bf16x3 kernels only): ptr is the bf16 HIGH plane of a split tensor (x = hi + lo, two bf16 per
// fp32), the LOW plane starts plane_stride bf16 elements later; strides and ld then count bf16 elements.
// `fmt` (single-product modes, input operands): CSN_FMT_F32 = fp32 words (split while staged); CSN_FMT_16 = a 16-bit map in the
// product's own type (bf16, or fp16 in math mode 3), staged by copy; CSN_FMT_F16_TO_BF16 = fp16 bits feeding a bf16 product
// (a forward tensor of math mode 3 read by its bf16 backward), converted in registers while staged.  Strides and ld count
// ELEMENTS of the operand's format.  An OUTPUT is a 16-bit map when planes == 1 in a one-plane mode.
enum { CSN_FMT_F32 = 0, CSN_FMT_16 = 1, CSN_FMT_F16_TO_BF16 = 2 };
struct CsnOperand {
  float* ptr;
  long long s0, s1, s2;
  const int* idx2;
  int ld;
  int planes;
  long long plane_stride;
  int fmt = CSN_FMT_F32;
};

// z2 is first mapped through the launch's evaluation list (if any), then through the operand's own slot map
CSN_DEVINL float* csn_operand_base(const CsnOperand& o, int z0, int z1, int z2) {
  long long i2 = o.idx2 ? (long long)o.idx2[z2] : (long long)z2;
  return o.ptr + o.s0 * z0 + o.s1 * z1 + o.s2 * i2;
}

// ---- dropout masks ------------------------------------------------------------------------------------
// Counter-based: the keep/drop decision of an element is a pure function of (seed, position), so the backward pass
// regenerates the forward's mask instead of storing it.  The mixer is the murmur3 32-bit finaliser.
// (tests/dropout_ref.py restates these functions in numpy; keep the two in step.)
CSN_DEVINL unsigned csn_mix32(unsigned h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}

// Attention-probability masks (the bulk of all mask decisions: T*T per block): a score block (evaluation, head, block)
// draws a 32-bit salt from (seed, block id) with two mixer rounds — wave-uniform, scalar unit — and ONE mixer round over (pair index ^ salt) decides two elements at once: the
// keys 2w and 2w+1 of query q have pair index w * max(score pitch, queries per block) + q and take the low / high 16 bits;
// keep  <=>  16-bit field >= p * 2^16.
CSN_DEVINL unsigned csn_block_salt(unsigned long long block_id, unsigned long long seed) {
  const unsigned h = csn_mix32((unsigned)block_id ^ (unsigned)seed);
  return csn_mix32(h + ((unsigned)(block_id >> 32) ^ (unsigned)(seed >> 32)) + 0x9e3779b9u);
}
CSN_DEVINL unsigned csn_pair_hash(unsigned pair_index, unsigned salt) { return csn_mix32(pair_index ^ salt); }
CSN_DEVINL unsigned csn_drop_threshold16(float p) { return (unsigned)(p * 65536.0f); }
// The fc-output masks (dropout before the residual add, csa_models.py:115-116) take the same form: an evaluation draws its
// salt from (seed, evaluation index) and one mixer round over (pair index ^ salt) decides the channels 2w (low 16 bits) and
// 2w+1 (high 16 bits) of point n, pair index = w * row pitch + n.  (A two-round hash per element cost the 256 x 256
// out-projection tile 36 k of its 100 k cycles: two quarter-rate multiplies per round.)
CSN_DEVINL unsigned csn_fc_pair(int channel, unsigned ld, unsigned n, unsigned salt) {
  return csn_pair_hash((unsigned)(channel >> 1) * ld + n, salt);
}
CSN_DEVINL bool csn_keep16(unsigned h, int odd, unsigned thr16) { return (odd ? (h >> 16) : (h & 0xffffu)) >= thr16; }

// ---- 16-bit matrix-core arithmetic: math modes 1..3 -----------------------------------------------------------------
// A mode is a compile-time policy of the 16-bit kernels (gemm_bf16x3.hip, attn_bf16x3.hip, outproj_ln.hip):
//   Bf16x3 (mode 1)  x = hi + lo (two bf16), product = hi*hi + hi*lo + lo*hi: NT = 3 matrix instructions, NPL = 2 planes
//   Bf16   (mode 2)  x = bf16(x), one product, one plane
//   F16    (mode 3)  x = fp16(x), one product, one plane (forward kernels only: gradients underflow fp16)
// 16-bit data is carried as raw shorts (LDS images, fragments, tile planes) and reinterpreted at the matrix instruction.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 csn_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 csn_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 csn_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 csn_f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4m __attribute__((ext_vector_type(4)));

namespace csn_mode {
struct Bf16x3 { static constexpr int NT = 3, NPL = 2; static constexpr bool HALF = false; };
struct Bf16 { static constexpr int NT = 1, NPL = 1; static constexpr bool HALF = false; };
struct F16 { static constexpr int NT = 1, NPL = 1; static constexpr bool HALF = true; };

template <bool H16> CSN_DEVINL short to16(float v) {
  if constexpr (H16) return __builtin_bit_cast(short, (_Float16)v);
  else return __builtin_bit_cast(short, (__bf16)v);
}
template <bool H16> CSN_DEVINL float from16(short v) {
  if constexpr (H16) return (float)__builtin_bit_cast(_Float16, v);
  else return (float)__builtin_bit_cast(__bf16, v);
}
template <bool H16> CSN_DEVINL s16x4 to16x4(const f32x4 v) {
  if constexpr (H16) return __builtin_bit_cast(s16x4, __builtin_convertvector(v, csn_f16x4));
  else return __builtin_bit_cast(s16x4, __builtin_convertvector(v, csn_bf16x4));
}
template <bool H16> CSN_DEVINL f32x4 from16x4(const s16x4 v) {
  if constexpr (H16) return __builtin_convertvector(__builtin_bit_cast(csn_f16x4, v), f32x4);
  else return __builtin_convertvector(__builtin_bit_cast(csn_bf16x4, v), f32x4);
}
// 4 floats -> 4 hi (round to nearest even) and, in the three-product mode, 4 lo = round(v - hi)
template <typename PR> CSN_DEVINL void split4(const f32x4 v, s16x4& hi, s16x4& lo) {
  hi = to16x4<PR::HALF>(v);
  if constexpr (PR::NT == 3) lo = to16x4<PR::HALF>(v - from16x4<PR::HALF>(hi));
  else lo = hi;
}
template <bool H16> CSN_DEVINL f32x16 mfma32(s16x8 a, s16x8 b, f32x16 c) {          // v_mfma_f32_32x32x16_{bf16,f16}
  if constexpr (H16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(csn_f16x8, a), __builtin_bit_cast(csn_f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(csn_bf16x8, a), __builtin_bit_cast(csn_bf16x8, b), c, 0, 0, 0);
}
template <bool H16> CSN_DEVINL f32x4m mfma16(s16x8 a, s16x8 b, f32x4m c) {           // v_mfma_f32_16x16x32_{bf16,f16}
  if constexpr (H16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(csn_f16x8, a), __builtin_bit_cast(csn_f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(csn_bf16x8, a), __builtin_bit_cast(csn_bf16x8, b), c, 0, 0, 0);
}
CSN_DEVINL s16x8 join8(s16x4 a, s16x4 b) { return s16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }
// 16-bit activation maps: 8 (4) fp16 values -> the same values as bf16 (round to nearest even); 4 values of either type -> fp32
CSN_DEVINL f32x4 f16x8_to_bf16x8(const f32x4 v) {
  s16x8 s = __builtin_bit_cast(s16x8, v);
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = to16<false>(from16<true>(s[j]));
  return __builtin_bit_cast(f32x4, s);
}
CSN_DEVINL s16x4 f16x4_to_bf16x4(const s16x4 v) { return to16x4<false>(from16x4<true>(v)); }
CSN_DEVINL f32x4 act16_to_f32(const s16x4 v, int fmt /* 1: bf16, 2: fp16 */) { return fmt == 2 ? from16x4<true>(v) : from16x4<false>(v); }
}  // namespace csn_mode

// sum / max across the two 32-lane halves of a wave (lane l <-> lane l ^ 32)
CSN_DEVINL float csn_xhalf(float v) { return __shfl_xor(v, 32, 64); }
