// shared by conv.hip: LDS swizzle for [rows][32-float] tiles read with ds_read_b128
#pragma once
// 16-byte chunk c (0..7) of row r lives at chunk c ^ ((r >> 1) & 7): the 16 lanes of one
// ds_read_b128 lane group (rows distinct mod 16, same chunk) cover all 16 slots of a bank row.
__device__ __forceinline__ int cv_off(int row, int chunk) {
  return row * 32 + ((chunk ^ ((row >> 1) & 7)) << 2);
}

// the same for a k-tile of KD floats per row (KD = 32: the function above; KD = 16: 64-byte rows, 4 chunks, and the chunk is
// XORed with bits 2-3 of the row, so that the 16 rows of a ds_read_b128 lane group — distinct in (row & 3, (row >> 2) & 3) —
// cover all 16 slots of a 256-byte bank row)
template <int KD>
__device__ __forceinline__ int lds_swz(int row) { return KD == 32 ? ((row >> 1) & 7) : ((row >> 2) & 3); }
template <int KD>
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * KD + ((chunk ^ lds_swz<KD>(row)) << 2); }
