// round 6: what v_permlane16_swap / v_permlane32_swap / DPP row_ror:8 return when both operands hold the same per-lane value
//   hipcc --offload-arch=gfx950 -O3 scripts/r6/lane_pair_check.hip -o /tmp/lane_pair_check && /tmp/lane_pair_check
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
  const unsigned x = threadIdx.x;
  const auto r16 = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  const auto r32 = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  const int d8 = __builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, false);
  out[threadIdx.x * 5 + 0] = r16[0]; out[threadIdx.x * 5 + 1] = r16[1];
  out[threadIdx.x * 5 + 2] = r32[0]; out[threadIdx.x * 5 + 3] = r32[1];
  out[threadIdx.x * 5 + 4] = d8;
}
int main() {
  int* d; int h[64 * 5];
  hipMalloc(&d, sizeof(h));
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 5) printf("lane %2d: swap16 -> (%2d, %2d)  swap32 -> (%2d, %2d)  row_ror:8 -> %2d\n", l, h[l * 5], h[l * 5 + 1], h[l * 5 + 2], h[l * 5 + 3], h[l * 5 + 4]);
  return 0;
}
