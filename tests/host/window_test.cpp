// Host model of a k-contiguous 16-bit operand fetch (csn_amd/csrc/csn_window.h): for every geometry, walk every 16-byte unit a
// kernel would request, apply the hardware's range check (dwords at or beyond the window read as zero and touch no memory) and
// the register rule (upper half cleared when it lies beyond the contraction), and check
//   (1) no byte outside the allocation is ever touched — the allocation ends EXACTLY with the last row's K elements;
//   (2) what reaches the product is the operand's value for k < K and zero for k >= K.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../csn_amd/csrc/csn_window.h"

static int check(int rows, int ld, int K, int tile_rows, int slab) {
  // allocation: rows of pitch ld, the last one cut after its K elements (nothing behind it)
  const long long alloc_el = (long long)(rows - 1) * ld + K;
  std::vector<unsigned short> mem(alloc_el);
  for (long long i = 0; i < alloc_el; ++i) mem[i] = (unsigned short)(1 + (i % 65000));
  int bad = 0;
  for (int m0 = 0; m0 < rows; m0 += tile_rows) {
    const int rows_valid = rows - m0 < tile_rows ? rows - m0 : tile_rows;
#ifdef CSN_TEST_FULL_ROW_WINDOW
    const long long win = (long long)rows_valid * ld * 2;                        // the window of the code that faulted: whole rows
#else
    const long long win = csn_kwin_bytes(rows_valid, ld, K, 2);                  // window of this row tile, from its first row
#endif
    const long long base = (long long)m0 * ld;                                   // element offset of the tile's first row
    for (int r = 0; r < tile_rows; ++r) {
      if (m0 + r >= rows) continue;                                              // (rows beyond M are switched off by the kernel)
      for (int k0 = 0; k0 < (K + slab - 1) / slab * slab; k0 += 8) {
        if (!csn_unit_starts_inside(k0, K)) continue;
        unsigned short unit[8];
        for (int d = 0; d < 4; ++d) {                                            // four dwords, each checked against the window
          const long long byte = ((long long)r * ld + k0) * 2 + 4 * d;
          if (byte + 4 <= win) {
            const long long el = base + (long long)r * ld + k0 + 2 * d;
            if (el < 0 || el + 1 >= alloc_el) { ++bad; std::printf("touch outside allocation: rows %d ld %d K %d m0 %d r %d k0 %d d %d\n", rows, ld, K, m0, r, k0, d); continue; }
            unit[2 * d] = mem[el];
            unit[2 * d + 1] = mem[el + 1];
          } else {
            unit[2 * d] = unit[2 * d + 1] = 0;
          }
        }
        if (csn_unit_upper_half_beyond(k0, K)) unit[4] = unit[5] = unit[6] = unit[7] = 0;
        for (int j = 0; j < 8; ++j) {
          const int k = k0 + j;
          const unsigned short want = k < K ? mem[base + (long long)r * ld + k] : 0;
          if (unit[j] != want) { ++bad; std::printf("wrong value: rows %d ld %d K %d m0 %d r %d k %d\n", rows, ld, K, m0, r, k); }
        }
      }
    }
  }
  return bad;
}

int main() {
  int bad = 0, cases = 0;
  const int lds[] = {500, 504, 1000, 1004, 10000, 36};
  for (int ld : lds)
    for (int K = 4; K <= ld && K <= 1004; K += (K < 40 ? 4 : 124))               // K % 8 in {0, 4}, K == ld included below
      for (int rows : {1, 3, 256, 260})
        for (int tile_rows : {128, 256}) { bad += check(rows, ld, K, tile_rows, 32); ++cases; }
  for (int ld : lds)
    for (int rows : {1, 256}) { bad += check(rows, ld, ld, 256, 32); ++cases; }   // the contraction fills the row: blocks that end the buffer
  std::printf("%d geometries, %d violations\n", cases, bad);
  return bad ? 1 : 0;
}
