// sustained fp32 MFMA rate on this device (registers only): the clock-adjusted ceiling for conv_gemm / km_assign
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int g = 0; g < 16; ++g) acc[i][g] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    a += 1e-6f;
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int g = 0; g < 16; ++g) s += acc[i][g];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* d; hipMalloc(&d, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256, 512, 1024, 2048}) {
    const int iters = 20000;
    k<<<blocks, 256>>>(d, 100, 0.5f, 0.25f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<blocks, 256>>>(d, iters, 0.5f, 0.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 /*waves*/ * iters * 16 * 4096.0;
    printf("blocks %5d: %.2f ms  %.1f TFLOP/s\n", blocks, ms, flop / ms / 1e9);
  }
  return 0;
}
