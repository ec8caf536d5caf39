// What does occupancy cost the fp32 matrix pipe?  Pure v_mfma_f32_32x32x2_f32 loops (operands in registers), swept over
//   W    = workgroups (of 4 waves = one wave per SIMD) resident per CU  -> waves per SIMD
//   NACC = independent accumulators per wave (1: every MFMA depends on the previous one)
//   BAR  = MFMAs between two workgroup barriers (0 = no barrier)
//   PRIO = 0 none | 1 s_setprio(1) around each burst
// Occupancy is pinned with dynamic LDS (160 KiB / W per workgroup) and a grid of exactly 256 * W workgroups.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_occ mfma_occ.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int BAR, int PRIO>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  extern __shared__ float lds[];
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  float a = 0.001f * (threadIdx.x & 7), b = 0.002f * (threadIdx.x & 3);
  for (int it = 0; it < iters; ++it) {
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int t = 0; t < 64 / NACC; ++t)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    if (BAR) __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
    for (int q = 0; q < 16; ++q) s += acc[i][q];
  out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x & 3];
}

template <int NACC, int BAR, int PRIO>
void run(float* out, int W) {
  const int iters = 4000;
  const int blocks = 256 * W;
  size_t lds = (160 * 1024 / W) & ~255;
  if (lds > 160 * 1024 - 256) lds = 160 * 1024 - 256;
  hipFuncSetAttribute((const void*)k<NACC, BAR, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) k<NACC, BAR, PRIO><<<blocks, 256, lds>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC, BAR, PRIO><<<blocks, 256, lds>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)blocks * 4 * iters * 64 * 4096.0;
  printf("W=%d nacc=%d bar=%d prio=%d : %7.2f ms %6.1f TFLOP/s\n", W, NACC, BAR, PRIO, ms, flop / ms / 1e9);
}

int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  for (int W = 1; W <= 6; ++W) {
    run<1, 0, 0>(out, W);
    run<2, 0, 0>(out, W);
    run<4, 0, 0>(out, W);
    run<1, 1, 0>(out, W);
    run<4, 1, 0>(out, W);
    run<4, 1, 1>(out, W);
  }
  return 0;
}
