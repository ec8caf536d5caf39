// shared by conv.hip: LDS swizzle for [rows][32-float] tiles read with ds_read_b128
#pragma once
// 16-byte chunk c (0..7) of row r lives at chunk c ^ ((r >> 1) & 7): the 16 lanes of one
// ds_read_b128 lane group (rows distinct mod 16, same chunk) cover all 16 slots of a bank row.
__device__ __forceinline__ int cv_off(int row, int chunk) {
  return row * 32 + ((chunk ^ ((row >> 1) & 7)) << 2);
}
