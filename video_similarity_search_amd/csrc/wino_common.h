// shared by conv.hip and conv_wino2.hip: packed-fp32 helpers and the F(4, 3) input transform
#pragma once
#include "common.h"

// Packed fp32 VALU ops as inline assembly: measured on this kernel family, wave time = MFMA cycles + VALU cycles (a diagnostic build
// with no memory traffic and no barrier ran the forward kernel at 74 % of the matrix pipe with ~130 VALU instructions per stage,
// the weight gradient at 66 % with ~46 per k-step — both what 64 cycles per MFMA plus 4 per VALU instruction predict), and left to
// itself the compiler scalarises the transforms (it schedules element j of every point towards MFMA j).  Two floats per instruction:
typedef float f32x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {        // a * b + c
  f32x2 d;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c));
  return d;
}
static __device__ __forceinline__ f32x2 pk_fnma(f32x2 a, f32x2 b, f32x2 c) {       // c - a * b
  f32x2 d;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "s"(b), "v"(c));
  return d;
}
static __device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 d;
  asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
static __device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {                 // a - b
  f32x2 d;
  asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// Two LDS words 256-byte units apart into ONE register pair (the compiler pairs neighbouring loads its own way and then moves
// registers around to build the pairs the packed ops need).  The compiler does not track the result of an inline-assembly load:
// the reader issues `s_waitcnt lgkmcnt(0)` itself before the first use.
template <int O0, int O1>
static __device__ __forceinline__ f32x2 lds_read2st64(unsigned addr) {
  static_assert(O0 >= 0 && O0 < 256 && O1 >= 0 && O1 < 256, "ds_read2st64_b32 offsets are 8 bits of 256-byte units");
  f32x2 d;
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(d) : "v"(addr), "n"(O0), "n"(O1));
  return d;
}
// V = B^T d for two floats at a time,  B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
// (c2 / c4 / c5: the constants 2, 4, 5 in both halves of a scalar register pair): 12 instructions for 12 outputs
static __device__ __forceinline__ void wino_bt6(const f32x2 (&d)[6], f32x2 (&V)[6], f32x2 c2, f32x2 c4, f32x2 c5) {
  const f32x2 t1 = pk_fnma(d[2], c4, d[4]), t2 = pk_fnma(d[1], c4, d[3]);
  const f32x2 t3 = pk_sub(d[4], d[2]), u = pk_sub(d[3], d[1]);
  V[0] = pk_fma(d[0], c4, pk_fnma(d[2], c5, d[4]));
  V[1] = pk_add(t1, t2);
  V[2] = pk_sub(t1, t2);
  V[3] = pk_fma(u, c2, t3);
  V[4] = pk_fnma(u, c2, t3);
  V[5] = pk_fma(d[1], c4, pk_fnma(d[3], c5, d[5]));
}

