// Do a wave's vector instructions hide under fp32 MFMAs on gfx950?  
//
// A stream of v_mfma_f32_32x32x2_f32 (64 cycles each on one SIMD, four rotating accumulators) with NF filler instructions of one
// kind behind every MFMA ("spread"), or GRP MFMAs followed by NF * GRP fillers in one block ("bunched": the shape of the Winograd
// kernels' transform block), at one and two waves per SIMD; and the cross-wave case: in a 512-thread workgroup waves 0-3 issue only
// MFMAs and waves 4-7 (their SIMD partners) only fillers.
//   cycles per MFMA = s_memtime over the loop / MFMAs of the wave (median over waves); wall TFLOP/s from HIP events beside it.
//   64.0 at one wave per SIMD (128.0 per wave at two) = the pipe's floor: everything above it is vector time that did NOT hide.
// Fillers run on eight independent registers (no dependency stalls) and are `asm volatile`; a sched_barrier pins them to their gap.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_coexec mfma_valu_coexec.hip ; run: ./mfma_valu_coexec > table.txt
// PMC: rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -- ./mfma_valu_coexec pmc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum Fill { NONE = 0, FMA, ADD, PKFMA, PKADD, MOV, IADD, SNOP, LDSRD, PKMUL, DMA_OOB, DMA_L2, GLD_OOB, GLD_L2, DSW128, DMA_NOV, DMA_FIXV };
static const char* fill_name[] = {"none", "v_fma_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_mov_b32", "v_add_u32", "s_nop 0",
                                  "ds_read_b128", "v_pk_mul_f32", "lds-dma oob", "lds-dma L2", "bufload oob", "bufload L2", "ds_write_b128", "lds-dma no-vaddr", "lds-dma fixed-vaddr"};

struct Regs {
  float f[8];
  f32x2 p[8];
  f32x4 q[4];
  unsigned u[8];
  unsigned lds_addr;
  __amdgpu_buffer_rsrc_t rsrc;
  float* ldsp;
};

template <int F>
static __device__ __forceinline__ void filler(Regs& R, int i) {
  const int j = i & 7;
  if constexpr (F == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(R.f[j]) : "v"(R.f[(j + 1) & 7]), "v"(R.f[(j + 2) & 7]));
  if constexpr (F == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(R.f[j]) : "v"(R.f[(j + 1) & 7]));
  if constexpr (F == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(R.p[j]) : "v"(R.p[(j + 1) & 7]), "v"(R.p[(j + 2) & 7]));
  if constexpr (F == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(R.p[j]) : "v"(R.p[(j + 1) & 7]));
  if constexpr (F == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(R.p[j]) : "v"(R.p[(j + 1) & 7]));
  if constexpr (F == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(R.u[j]) : "v"(R.u[(j + 1) & 7]));
  if constexpr (F == IADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(R.u[j]) : "v"(R.u[(j + 1) & 7]));
  if constexpr (F == SNOP) asm volatile("s_nop 0");
  if constexpr (F == LDSRD) asm volatile("ds_read_b128 %0, %1" : "=v"(R.q[i & 3]) : "v"(R.lds_addr));
  if constexpr (F == DSW128) asm volatile("ds_write_b128 %0, %1" ::"v"(R.lds_addr + 4096u * (i & 3)), "v"(R.q[i & 3]));
  // LDS-DMA (buffer_load_dwordx4 ... lds): 1 KiB per wave-instruction into the wave's own LDS area; OOB = out of range (zeros, no
  // memory traffic), L2 = a 64 KiB buffer every wave re-reads
  if constexpr (F == DMA_OOB || F == DMA_L2) {
    const unsigned voff = F == DMA_OOB ? 0xFFFFFF00u : R.lds_addr + 1024u * (unsigned)(i & 31);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(R.rsrc, (__attribute__((address_space(3))) void*)(R.ldsp + 256 * (i & 7)), 16, (int)voff, 0, 0, 0);
  }
  // the same LDS-DMA with NO vector address operand (voffset = off: every lane reads the same 16 bytes; the scalar offset walks) and with a
  // loop-invariant vector address (no VALU instruction per piece): is it the vector register read that holds the matrix pipe?
  if constexpr (F == DMA_NOV)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(R.rsrc, (__attribute__((address_space(3))) void*)(R.ldsp + 256 * (i & 7)), 16, 0, (int)(1024u * (unsigned)(i & 31)), 0, 0);
  if constexpr (F == DMA_FIXV)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(R.rsrc, (__attribute__((address_space(3))) void*)(R.ldsp + 256 * (i & 7)), 16, (int)R.lds_addr, (int)(1024u * (unsigned)(i & 31)), 0, 0);
  if constexpr (F == GLD_OOB || F == GLD_L2) {
    const unsigned voff = F == GLD_OOB ? 0xFFFFFF00u : R.lds_addr + 1024u * (unsigned)(i & 31);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(R.rsrc, (int)voff, 0, 0);
    asm volatile("" ::"v"(v));          // keeps the load, waits for it (the compiler's vmcnt): a register-staged load's cost
  }
}

// MODE 0: spread (NF fillers behind every MFMA); 1: bunched (GRP MFMAs, then NF * GRP fillers); 2: cross-wave (waves 0-3 MFMAs only,
// waves 4-7 fillers only: NF fillers per "MFMA slot" of 64 cycles they would like to run beside)
template <int F, int NF, int MODE, int GRP>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, const float* src) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  Regs R;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    R.f[i] = 1.0f + 1e-7f * (threadIdx.x + i);
    R.p[i] = (f32x2){1.0f + 1e-7f * i, 1.0f - 1e-7f * i};
    R.u[i] = threadIdx.x + i;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) R.q[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  R.lds_addr = (threadIdx.x & 63) * 16;
  R.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 65536, 0x00020000);
  R.ldsp = lds + 4096 + (threadIdx.x >> 6) * 2048;          // 8 KiB of LDS per wave for the DMA fills
  const float a = 0.001f * (threadIdx.x & 7), b = 0.002f * (threadIdx.x & 3);
  const bool mfma_role = MODE != 2 || wave < 4, fill_role = MODE != 2 || wave >= 4;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 3) {             // thin: one filler behind every NF-th MFMA (GRP / NF fillers per GRP MFMAs)
#pragma unroll
      for (int g = 0; g < GRP; ++g) {
        acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (g % NF == NF - 1) filler<F>(R, g / NF);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (F == DMA_OOB || F == DMA_L2 || F == DMA_NOV || F == DMA_FIXV) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE == 1) {
      if (mfma_role) {
#pragma unroll
        for (int g = 0; g < GRP; ++g) acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g & 3], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NF * GRP; ++n) filler<F>(R, n);
      if (F == LDSRD) asm volatile("s_waitcnt lgkmcnt(0)");
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int g = 0; g < GRP; ++g) {
        if (mfma_role) acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (fill_role) {
#pragma unroll
          for (int n = 0; n < NF; ++n) filler<F>(R, g * NF + n);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (F == LDSRD) asm volatile("s_waitcnt lgkmcnt(0)");
      if (F == DMA_OOB || F == DMA_L2 || F == DMA_NOV || F == DMA_FIXV) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    for (int q = 0; q < 16; ++q) s += acc[i][q];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += R.f[i] + R.p[i][0] + R.p[i][1] + (float)R.u[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += R.q[i][0] + R.q[i][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x & 3];
  if ((threadIdx.x & 63) == 0) cyc[(size_t)blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

static float* g_out;
static unsigned long long* g_cyc;
static bool g_pmc = false;
static float* g_src;

// WPS = waves per SIMD (1: 256-thread workgroups, one per CU; 2: 512-thread workgroups, one per CU — MODE 2 needs that shape)
template <int F, int NF, int MODE, int GRP = 16>
void run(int WPS) {
  const int threads = (MODE == 2 || WPS == 2) ? 512 : 256;
  const int blocks = 256;
  const int iters = g_pmc ? 400 : 2000;
  const size_t ldsb = 150 * 1024;                     // one workgroup per CU
  hipFuncSetAttribute((const void*)k<F, NF, MODE, GRP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  if (!g_pmc) k<F, NF, MODE, GRP><<<blocks, threads, ldsb>>>(g_out, g_cyc, iters, g_src);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<F, NF, MODE, GRP><<<blocks, threads, ldsb>>>(g_out, g_cyc, iters, g_src);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const int nw = blocks * threads / 64;
  std::vector<unsigned long long> c(nw);
  hipMemcpy(c.data(), g_cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  // the MFMA waves' cycles (MODE 2: waves 0-3 of each workgroup)
  std::vector<unsigned long long> m;
  for (int i = 0; i < nw; ++i)
    if (MODE != 2 || (i % 8) < 4) m.push_back(c[i]);
  std::sort(m.begin(), m.end());
  const double med = (double)m[m.size() / 2];
  double fmed = 0;
  if (MODE == 2 && NF > 0) {
    std::vector<unsigned long long> f;
    for (int i = 0; i < nw; ++i)
      if ((i % 8) >= 4) f.push_back(c[i]);
    std::sort(f.begin(), f.end());
    fmed = (double)f[f.size() / 2] / ((double)iters * GRP * NF);
  }
  const double mfmas_per_wave = (double)iters * GRP;
  const int mfma_waves_per_simd = MODE == 2 ? 1 : WPS;
  const double flop = (double)blocks * 4 * mfma_waves_per_simd * mfmas_per_wave * 4096.0;
  printf("%-8s %-13s nf=%2d grp=%2d waves/SIMD=%d : %7.1f cyc/MFMA/wave  %6.1f cyc/MFMA/SIMD  %6.1f TFLOP/s  (floor 64.0/SIMD)",
         MODE == 0 ? "spread" : MODE == 1 ? "bunched" : MODE == 2 ? "xwave" : "thin", fill_name[F], NF, GRP, MODE == 2 ? 2 : WPS, med / mfmas_per_wave,
         med / mfmas_per_wave / mfma_waves_per_simd, flop / ms / 1e9);
  if (MODE == 2 && NF > 0) printf("  filler wave: %.1f cyc/filler (whole kernel, incl. its time alone)", fmed);
  printf("\n");
  fflush(stdout);
}

int main(int argc, char** argv) {
  g_pmc = argc > 1 && !strcmp(argv[1], "pmc");
  hipMalloc(&g_out, 256 * 512 * 4);
  hipMalloc(&g_cyc, 256 * 8 * 8);
  hipMalloc(&g_src, 65536);
  hipMemset(g_src, 0, 65536);
  if (g_pmc) {
    // a few dispatches for a counter pass: kernel-trace order = this order
    run<NONE, 0, 0>(2);
    run<FMA, 8, 0>(2);
    run<PKFMA, 4, 0>(2);
    run<FMA, 2, 1, 24>(2);
    run<PKFMA, 1, 1, 24>(2);
    run<FMA, 8, 2>(2);
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "dma2")) {
    // is it the DMA's vector address operand that costs the matrix pipe?  the same piece with no vector address / a loop-invariant one
    for (int wps = 1; wps <= 2; ++wps) {
      run<NONE, 0, 0, 24>(wps);
      run<DMA_L2, 4, 3, 24>(wps);
      run<DMA_FIXV, 4, 3, 24>(wps);
      run<DMA_NOV, 4, 3, 24>(wps);
      run<DMA_L2, 2, 3, 24>(wps);
      run<DMA_FIXV, 2, 3, 24>(wps);
      run<DMA_NOV, 2, 3, 24>(wps);
    }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "dma")) {
    // memory-instruction issue cost beside fp32 MFMAs: the Winograd stage issues 6 LDS-DMA pieces per 24 MFMAs (thin nf = 4)
    for (int wps = 1; wps <= 2; ++wps) {
      run<NONE, 0, 0, 24>(wps);
      run<DMA_OOB, 4, 3, 24>(wps);
      run<DMA_OOB, 2, 3, 24>(wps);
      run<DMA_L2, 4, 3, 24>(wps);
      run<DMA_L2, 2, 3, 24>(wps);
      run<GLD_OOB, 4, 3, 24>(wps);
      run<GLD_L2, 4, 3, 24>(wps);
      run<GLD_L2, 2, 3, 24>(wps);
      run<DSW128, 4, 3, 24>(wps);
      run<DSW128, 2, 3, 24>(wps);
      run<LDSRD, 2, 3, 24>(wps);
      run<LDSRD, 1, 3, 24>(wps);
      run<DMA_L2, 1, 1, 6>(wps);       // bunched: 6 MFMAs, then 6 DMA pieces
    }
    run<DMA_OOB, 1, 2>(2);
    run<DMA_L2, 1, 2>(2);
    run<GLD_L2, 1, 2>(2);
    run<DSW128, 1, 2>(2);
    return 0;
  }
  for (int wps = 1; wps <= 2; ++wps) {
    run<NONE, 0, 0>(wps);
    run<FMA, 1, 0>(wps);
    run<FMA, 2, 0>(wps);
    run<FMA, 4, 0>(wps);
    run<FMA, 8, 0>(wps);
    run<FMA, 12, 0>(wps);
    run<FMA, 16, 0>(wps);
    run<ADD, 4, 0>(wps);
    run<ADD, 8, 0>(wps);
    run<PKFMA, 1, 0>(wps);
    run<PKFMA, 2, 0>(wps);
    run<PKFMA, 4, 0>(wps);
    run<PKFMA, 8, 0>(wps);
    run<PKADD, 2, 0>(wps);
    run<PKADD, 4, 0>(wps);
    run<PKMUL, 4, 0>(wps);
    run<MOV, 4, 0>(wps);
    run<MOV, 8, 0>(wps);
    run<IADD, 4, 0>(wps);
    run<IADD, 8, 0>(wps);
    run<SNOP, 8, 0>(wps);
    run<LDSRD, 1, 0>(wps);
    run<LDSRD, 2, 0>(wps);
    // the Winograd stage's shape: 24 MFMAs, then one block of vector work (the stage's 24 packed ops = nf 1; as 48 plain = nf 2)
    run<PKFMA, 1, 1, 24>(wps);
    run<FMA, 2, 1, 24>(wps);
    run<PKFMA, 2, 1, 24>(wps);
    run<FMA, 4, 1, 24>(wps);
    run<MOV, 2, 1, 24>(wps);
  }
  // across the two waves of a SIMD: MFMA-only wave beside a filler-only wave
  run<NONE, 0, 2>(2);
  run<FMA, 4, 2>(2);
  run<FMA, 8, 2>(2);
  run<FMA, 16, 2>(2);
  run<PKFMA, 4, 2>(2);
  run<PKFMA, 8, 2>(2);
  run<MOV, 8, 2>(2);
  run<IADD, 8, 2>(2);
  run<LDSRD, 2, 2>(2);
  return 0;
}
