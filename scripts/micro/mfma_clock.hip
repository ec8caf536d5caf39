// in-kernel clock under different instruction mixes (MI355X_MICROARCH.md 'DVFS give-back' item 6):
// clock = d(s_memtime) / d(s_memrealtime) * 100 MHz.  Mixes: MFMA only; MFMA + LDS reads; MFMA + LDS + streaming global loads.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ g, size_t gmask, float* out, unsigned long long* stamps, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)(i & 15) * 0.01f;
  __syncthreads();
  f32x16 acc0, acc1;
  for (int q = 0; q < 16; ++q) { acc0[q] = 0.f; acc1[q] = 0.f; }
  const int lane = threadIdx.x & 63;
  f32x4 a = {0.1f, 0.2f, 0.3f, 0.4f}, b = {0.5f, 0.6f, 0.7f, 0.8f};
  f32x4 gsum = {0.f, 0.f, 0.f, 0.f};
  size_t goff = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 1) {
      a = *(const f32x4*)&lds[((it * 64 + lane) * 4) & 8188];
      b = *(const f32x4*)&lds[((it * 64 + lane) * 4 + 4096) & 8188];
    }
    if (MODE >= 2) {
      gsum += *(const f32x4*)(g + (goff & gmask));
      goff += (size_t)gridDim.x * 1024;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[t], a[t], acc1, 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = gsum.x + gsum.y;
  for (int q = 0; q < 16; ++q) s += acc0[q] + acc1[q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}
template <int MODE>
void run(const char* name, const float* g, size_t gmask, float* out, unsigned long long* st, int blocks) {
  const int iters = 40000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) k<MODE><<<blocks, 256>>>(g, gmask, out, st, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(g, gmask, out, st, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 2);
  hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
  std::vector<double> clk;
  for (int b = 0; b < blocks; ++b) clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0);
  std::sort(clk.begin(), clk.end());
  double flop = (double)blocks * 4 * iters * 8 * 4096.0;
  printf("%-34s blocks %4d: %7.2f ms %6.1f TFLOP/s   in-kernel clock median %.0f MHz (min %.0f max %.0f)\n", name, blocks, ms,
         flop / ms / 1e9, clk[clk.size() / 2], clk.front(), clk.back());
}
int main() {
  float *g, *out; unsigned long long* st;
  size_t gbytes = (size_t)1 << 30;
  hipMalloc(&g, gbytes); hipMemset(g, 0, gbytes); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&st, 4096 * 16);
  size_t gmask = gbytes / 4 - 1;
  for (int blocks : {512, 1280}) {
    run<0>("MFMA only", g, gmask, out, st, blocks);
    run<1>("MFMA + 2 ds_read_b128 / 8 MFMA", g, gmask, out, st, blocks);
    run<2>("MFMA + LDS + 16B/lane global stream", g, gmask, out, st, blocks);
  }
  return 0;
}
