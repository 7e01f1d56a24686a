// Buffer-window arithmetic of the k-contiguous 16-bit operands (gemm_bf16x3.hip, AF / BF != 0), kept apart so that the host
// can test it (tests/host/window_test.cpp, run by tests/test_cpu_window.py).
//
// A 16-bit map operand is fetched in 16-byte units of 8 elements along the contraction index k.  Contraction lengths are
// multiples of 4, not of 8 (blocks of 500 queries; k chunks of a weight gradient), so the last unit of a row can straddle the
// end of the contraction: its lower half holds k < K, its upper half belongs to whatever lies behind — the next row, the next
// block, or, for the last row of the last item, NOTHING: the allocation ends there (found as a memory fault at config-3 size).
// Two rules make every fetch safe and every product exact:
//   * the operand's buffer window ends with the last valid row's K elements — dwords beyond a window are never fetched and read
//     as zero (hardware range check of the buffer instructions);
//   * a unit whose upper half lies at k >= K has that half cleared in registers (inside the window it holds real data of the
//     next row or block, which must not enter the product).
#pragma once
#ifndef CSN_HD
#ifdef __HIPCC__
#define CSN_HD __host__ __device__ __forceinline__
#else
#define CSN_HD inline
#endif
#endif

// bytes of the window over `rows_valid` rows of pitch `ld` elements of `es` bytes whose contraction runs over the first K
// elements of every row: the last row contributes its K elements only
CSN_HD long long csn_kwin_bytes(int rows_valid, int ld, int K, int es) {
  return rows_valid <= 0 ? 0 : ((long long)(rows_valid - 1) * ld + K) * es;
}
// 1: the upper half (elements 4..7) of the 16-byte unit that starts at contraction index k0 lies beyond the contraction
CSN_HD int csn_unit_upper_half_beyond(int k0, int K) { return k0 + 8 > K; }
// 1: the unit starts inside the contraction (units that start at or beyond K are not fetched at all)
CSN_HD int csn_unit_starts_inside(int k0, int K) { return k0 < K; }
