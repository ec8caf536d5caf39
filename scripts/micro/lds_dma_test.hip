// does buffer_load ... lds (16 B/lane) zero-fill LDS for out-of-range lanes on gfx950?  and the lane -> LDS address map
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float* src, unsigned bytes, const int* offs, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 256 + 64];
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  for (int i = threadIdx.x; i < 2 * 256 + 64; i += 64) lds[i] = -7.f;
  __syncthreads();
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 256), 16, offs[threadIdx.x], 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * 256 + 64; i += 64) out[i] = lds[i];
}
int main() {
  const int n = 1024;
  std::vector<float> h(n); for (int i = 0; i < n; ++i) h[i] = (float)i;
  std::vector<int> off(64);
  for (int l = 0; l < 64; ++l) off[l] = ((l * 7) % 61) * 16;      // arbitrary per-lane 16-B aligned source offsets
  off[5] = 0xFFFFFF00; off[40] = n * 4;                            // out of range lanes
  float *d, *o; int* doff;
  hipMalloc(&d, n * 4); hipMalloc(&o, 1024 * 4); hipMalloc(&doff, 64 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(doff, off.data(), 64 * 4, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, n * 4, doff, o);
  std::vector<float> r(576); hipMemcpy(r.data(), o, 576 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
    float exp = (l == 5 || l == 40) ? 0.f : (float)(off[l] / 4 + e);
    if (r[256 + l * 4 + e] != exp) { if (bad < 8) printf("lane %d e %d got %f exp %f\n", l, e, r[256 + l * 4 + e], exp); ++bad; }
  }
  for (int i = 0; i < 256; ++i) if (r[i] != -7.f) { ++bad; if (bad < 12) printf("guard before clobbered at %d: %f\n", i, r[i]); }
  for (int i = 512; i < 576; ++i) if (r[i] != -7.f) { ++bad; if (bad < 12) printf("guard after clobbered at %d: %f\n", i, r[i]); }
  printf("lds-dma test: %s (%d mismatches)\n", bad ? "FAIL" : "OK: linear lane*16 map, OOB lanes write zeros", bad);
  return bad != 0;
}
