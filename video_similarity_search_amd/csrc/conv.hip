// 3-D convolution on gfx950 as a table-driven gather-GEMM on exact-fp32 MFMA.
//
// Replaces the cuDNN/MIOpen calls behind nn.Conv3d in /root/reference/models/resnet.py:11-25,126-131
// (forward, and the autograd dgrad / wgrad), and nn.Linear (:182-184) as the 1x1x1 special case.
//
// Data layout: activations NDHWC fp32 ([B, T, H, W, C], C % 4 == 0), weights packed K-contiguous
// ([N][Kp], k = tap * C + c).  One kernel serves forward, data-gradient (stride-1 and each parity
// class of a stride-2 layer) and linear layers: a host-built table maps every 16-byte K-chunk to
// (source element delta, tap offsets for the bounds test, weight column).  Rows of the implicit
// GEMM are output positions, so a tile row is one pixel's channel run: 16-byte coalesced loads.
//
//   A (im2col rows, gathered)  [BM x 32]  -> LDS (XOR-swizzled 16-B chunks) -> MFMA A operand
//   B (packed weights)         [BN x 32]  -> LDS                           -> MFMA B operand
//   v_mfma_f32_32x32x2_f32, 64 cycles each: 1 ds_read_b128 feeds 4 MFMAs per operand tile, so the
//   matrix pipe, not LDS or address math, is the limiter.  Accumulation is exact fp32 (k-ordered
//   fma chain per accumulator); no reduced-precision path.
//
// Epilogue (fused, branch-free buffer ops): + bias, per-channel affine (eval-mode BN), + addend (residual /
// gradient accumulation), ReLU-backward mask of the consuming layer, ReLU, and per-workgroup BatchNorm
// partials written to a slab that bn.hip merges in fixed order: forward (sum, sum (v - mean_wg)^2) or
// backward (sum g, sum g * xhat).  Kernels: conv_gemm_dma_kernel (LDS-DMA ring; default), its multi-GEMM
// and split-K forms, conv_gemm_kernel (register-staged; stem), conv_wgrad_dma_kernel (+ row table).
#include "common.h"
#include "conv_common.h"
#include "conv_internal.h"
#include <type_traits>
#ifndef SLIC_WG_ILVQ
#define SLIC_WG_ILVQ 7   // same for the weight gradient's eight MFMA groups per tile
#endif
#ifndef SLIC_PRIO_EDGE
#define SLIC_PRIO_EDGE 3 // wave priority of the gather-GEMM's prologue and epilogue (the k loop runs at 0)
#endif
#ifndef SLIC_ILVQ
#define SLIC_ILVQ 2      // MFMA groups (of 4 per k-tile) over which the next tile's DMAs are spread (4 -> 2: +0.6 % on the step: the DMAs get half a tile more lead)
#endif
#include <stdlib.h>
#ifndef SLIC_GEMM0_WPE
#define SLIC_GEMM0_WPE 4   // waves per SIMD of the register-staged kernel, pinned: left to itself the compiler shuttled the accumulator between
#endif                     // VGPRs and AGPRs every k-tile (32 v_accvgpr moves per 32 MFMAs — vector instructions, which stop the matrix pipe)

// In-kernel time stamps (MI355X_MICROARCH.md, 'In-kernel stamps'): ONLY in the diagnostic build scripts/stamps_conv.py makes
// (-DSLIC_STAMPS, a separate library under csrc/_exp/); in the shipped library the macro is empty and no stamp executes.
#ifdef SLIC_STAMPS
__device__ unsigned long long* slic_stamps_buf = nullptr;     // [workgroups][8], memory no other code reads
extern "C" int slic_debug_set_stamps(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(slic_stamps_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#define SLIC_STAMP(wg, slot)                                                                              \
  do {                                                                                                   \
    if (threadIdx.x == 0 && slic_stamps_buf) slic_stamps_buf[(size_t)(wg) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define SLIC_STAMP_ID(wg)                                                                                 \
  do {                                                                                                   \
    if (threadIdx.x == 0 && slic_stamps_buf)                                                             \
      slic_stamps_buf[(size_t)(wg) * 8] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492); \
  } while (0)
#else
#define SLIC_STAMP(wg, slot)
#define SLIC_STAMP_ID(wg)
#endif

#include "conv_epilogue.h"

template <int BM, int BN, int WM, int WN, int TM, int TN>
__device__ __forceinline__ void conv_epilogue(const SlicConvArgs& p, f32x16 (&acc)[TM][TN], float* lds, int64_t m0, int n0,
                                              int wm, int wn, int r, int h, int tid) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  float* tile = lds;                       // [BM][BN]: acc + bias
  // ---- 1. registers -> LDS image (the k-loop's ring is dead: every caller has drained its DMAs and passed a barrier)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = wn * WTN + j * 32 + r;
    const int n = n0 + col;
    const float bias = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int row = wm * WTM + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
        tile[row * BN + col] = acc[i][j][g] + bias;
      }
  }
  __syncthreads();
  conv_epilogue_rows<BM, BN>(p, lds, m0, n0, tid);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SLIC_GEMM0_WPE, SLIC_GEMM0_WPE))) void conv_gemm_kernel(const SlicConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WTM = BM / WM, WTN = BN / WN;   // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;   // 32x32 MFMA tiles per wave
  constexpr int AL = BM / 32, BL = BN / 32;     // 16-byte chunks per thread per k-tile
  static_assert(WM * WN == 4, "4 waves");
  float* As = lds;                               // [2][BM*32]
  float* Bs = lds + 2 * BM * 32;                 // [2][BN*32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  // ---- staging setup: thread owns chunk column cq of rows srow + 32 i.
  // Loads are buffer loads (raw, stride 0): an out-of-range offset returns zeros, so padding taps, ragged rows
  // and ragged channels need no branch and no select — the bounds test just picks the offset.
  const int cq = tid & 7, srow = tid >> 3;
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wgt = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, 0, p.wgt_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFF00u;
  unsigned aoff[AL];    // byte offset of the row's base pixel
  unsigned rmask[AL];   // bit (7*dim + o + 3) set iff base coordinate + o is inside the source along dim
#pragma unroll
  for (int i = 0; i < AL; ++i) {
    const int64_t m = m0 + srow + 32 * i;
    aoff[i] = 0; rmask[i] = 0;
    if (m < p.M) {
      unsigned rr = (unsigned)m;                     // M < 2^31 (checked on the host)
      const int gc = (int)(rr % (unsigned)p.Gc); rr /= (unsigned)p.Gc;
      const int gb = (int)(rr % (unsigned)p.Gb); rr /= (unsigned)p.Gb;
      const int ga = (int)(rr % (unsigned)p.Ga); rr /= (unsigned)p.Ga;   // rr = batch
      const int a0 = ga * p.sa, b0 = gb * p.sb, c0 = gc * p.sc;
      aoff[i] = (((rr * p.Ts + a0) * p.Hs + b0) * p.Ws + c0) * (unsigned)p.Cs * 4u;
      unsigned mk = 0;
#pragma unroll
      for (int o = -3; o <= 3; ++o) {
        mk |= ((unsigned)(a0 + o) < (unsigned)p.Ts ? 1u : 0u) << (o + 3);
        mk |= ((unsigned)(b0 + o) < (unsigned)p.Hs ? 1u : 0u) << (7 + o + 3);
        mk |= ((unsigned)(c0 + o) < (unsigned)p.Ws ? 1u : 0u) << (14 + o + 3);
      }
      rmask[i] = mk;
    }
  }
  unsigned woff[BL];
#pragma unroll
  for (int i = 0; i < BL; ++i) {
    const int n = n0 + srow + 32 * i;
    woff[i] = n < p.N ? (unsigned)n * (unsigned)p.ldw * 4u : OOB;
  }
  // two register staging sets: tile kt+2 is in flight while tile kt is computed and tile kt+1 is written to LDS,
  // so a load has a whole k-tile (plus the other workgroups' tiles) to come back from L2 / Infinity Cache / HBM
  i32x4 gA[2][AL], gB[2][BL];
  const int nk = p.nchunks >> 3;
  int4 e_next = nk > 0 ? ((const int4*)p.tab)[cq] : make_int4(0, -1, 0, 0);   // table entry one tile ahead
  auto gload = [&](int kt, i32x4 (&ga)[AL], i32x4 (&gb)[BL]) {
    const int4 e = e_next;
    const unsigned tm = (unsigned)e.y;             // 0xFFFFFFFF for an all-zero chunk: never a subset of rmask
    const unsigned dlt = (unsigned)e.x * 4u;
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const unsigned off = ((rmask[i] & tm) == tm) ? aoff[i] + dlt : OOB;
      ga[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, (int)off, 0, 0);
    }
    const unsigned wc = e.y == -1 ? OOB : (unsigned)e.z * 4u;
#pragma unroll
    for (int i = 0; i < BL; ++i) {
      const unsigned off = (woff[i] == OOB || wc == OOB) ? OOB : woff[i] + wc;
      gb[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_wgt, (int)off, 0, 0);
    }
    if (kt + 1 < nk) e_next = ((const int4*)p.tab)[(kt + 1) * 8 + cq];
  };
  auto lwrite = [&](int buf, i32x4 (&ga)[AL], i32x4 (&gb)[BL]) {
#pragma unroll
    for (int i = 0; i < AL; ++i) *(i32x4*)&As[buf * BM * 32 + cv_off(srow + 32 * i, cq)] = ga[i];
#pragma unroll
    for (int i = 0; i < BL; ++i) *(i32x4*)&Bs[buf * BN * 32 + cv_off(srow + 32 * i, cq)] = gb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

  const int r = lane & 31, h = lane >> 5;
  auto compute = [&](int buf) {
    const float* Ab = As + buf * BM * 32;
    const float* Bb = Bs + buf * BN * 32;
    f32x4 a[2][TM], b[2][TN];     // LDS operands double-buffered in registers: group q+1 is in flight under q's MFMAs
#pragma unroll
    for (int i = 0; i < TM; ++i) a[0][i] = *(const f32x4*)&Ab[cv_off(wm * WTM + i * 32 + r, h)];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[0][j] = *(const f32x4*)&Bb[cv_off(wn * WTN + j * 32 + r, h)];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cur = q & 1, nxt = cur ^ 1;
      if (q < 3) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[nxt][i] = *(const f32x4*)&Ab[cv_off(wm * WTM + i * 32 + r, 2 * (q + 1) + h)];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[nxt][j] = *(const f32x4*)&Bb[cv_off(wn * WTN + j * 32 + r, 2 * (q + 1) + h)];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][t], b[cur][j][t], acc[i][j], 0, 0, 0);
      // issue order: next group's LDS reads first, then this group's MFMAs (reads land under 4*TM*TN*64 cycles)
      if (q < 3) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);
    }
  };
  if (nk > 0) {
    gload(0, gA[0], gB[0]);
    lwrite(0, gA[0], gB[0]);
    if (nk > 1) gload(1, gA[1], gB[1]);
  }
  __syncthreads();
  for (int kt0 = 0; kt0 < nk; kt0 += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {       // unrolled: register sets and LDS buffers are compile-time
      const int kt = kt0 + half;
      if (kt < nk) {
        // tile kt sits in LDS buffer `half`; set `half` is free, set `half ^ 1` carries tile kt + 1
        if (kt + 2 < nk) gload(kt + 2, gA[half], gB[half]);
        compute(half);
        if (kt + 1 < nk) lwrite(half ^ 1, gA[half ^ 1], gB[half ^ 1]);
        __syncthreads();
      }
    }
  }

  conv_epilogue<BM, BN, WM, WN, TM, TN>(p, acc, lds, m0, n0, wm, wn, r, h, tid);
}

// ------------------------------------------------------------------------------------------
// LDS-DMA variant (source channels % 32 == 0, i.e. every R3D-18 layer but the stem): a 32-wide K tile never
// straddles a tap, so tap offset / mask / weight base are wave-uniform per k-tile and come from a per-TAP table
// through scalar loads; operands go global -> LDS directly (buffer_load_dwordx4 ... lds, 1 KB per wave-instruction,
// out-of-range lanes write zeros), no VGPR staging, no ds_write, and a ring of LDS stages with counted vmcnt keeps
// two k-tiles in flight across ONE raw barrier per tile.  LDS image = the same [rows][32] swizzled layout: the
// DMA writes lane-linear, so the swizzle is applied to the per-lane SOURCE chunk (p ^ ((row >> 1) & 7)).
// ------------------------------------------------------------------------------------------
// waves per SIMD the LDS footprint allows (one wave of each resident workgroup per SIMD): the register allocator is held to
// that occupancy, or the epilogue's batched loads would cost the main loop a workgroup per CU
constexpr int conv_dma_waves(int BM, int BN, int STAGES, int KD = 32) {
  const int ring = STAGES * (BM + BN) * KD * 4, epi = conv_epi_lds_floats(BM, BN) * 4;
  const int lds = ring > epi ? ring : epi;
  const int w = (160 * 1024) / lds;
  return w > 6 ? 6 : (w < 1 ? 1 : w);
}

// The kernel body, shared by the one-GEMM launch and the multi-GEMM launch (several SlicConvArgs in one grid: the parity
// classes of a stride-2 data gradient).  (bxi, gdx, byi, bzi) stand for (blockIdx.x, gridDim.x, blockIdx.y, blockIdx.z).
template <int BM, int BN, int WM, int WN, int STAGES, int KD>
__device__ __forceinline__ void conv_gemm_dma_body(const SlicConvArgs& p, const int xcd_remap, float* __restrict__ slab,
                                                   const int kt_per_split, float* lds, const int bxi, const int gdx,
                                                   const int byi, const int bzi, const int64_t slab_zstride) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(KD == 32 || KD == 16, "k-tile depth");
  constexpr int CPW = KD / 4;                                 // 16-byte chunks per LDS row
  constexpr int RALL = 256 / CPW;                             // rows the 256 threads cover with one DMA instruction per wave
  constexpr int AL = BM / RALL, BL = BN / RALL;               // DMA instructions per wave per k-tile
  constexpr int NG = KD / 8;                                  // MFMA groups per k-tile (one ds_read_b128 per operand tile each)
  constexpr int TILE_FLOATS = (BM + BN) * KD;                 // one k-tile (KD deep) of A and B
  static_assert(BM % RALL == 0 && BN % RALL == 0, "tile rows per DMA pass");
  constexpr int STAGE_FLOATS = TILE_FLOATS;                   // a ring stage holds one k-tile: one barrier per k-tile
  static_assert(WM * WN == 4, "4 waves");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the DMAs' LDS bases (M0) then need no v_readfirstlane each
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware order: consecutive workgroup ids run on different XCDs (id % 8), so hand each XCD a contiguous range of
  // row blocks — neighbouring rows (the taps' halo) are then re-read through the same L2
  int mb = bxi;
  if (xcd_remap) mb = (bxi & 7) * (gdx >> 3) + (bxi >> 3);
  const int64_t m0 = (int64_t)mb * BM;
  if (m0 >= p.M) return;
  const int n0 = byi * BN;
  [[maybe_unused]] const int stamp_wg = bxi + gdx * byi;
  SLIC_STAMP_ID(stamp_wg);
  SLIC_STAMP(stamp_wg, 1);
  // A workgroup arrives on a CU whose other resident workgroups sit in their k loops: their waves are older and always have an
  // MFMA waiting for the matrix pipe, and vector issue goes to the oldest wave first — measured with in-kernel stamps
  // (scripts/stamps_conv.py), this prologue's few hundred VALU instructions then take 40-60 us (the k loop itself: ~120 us),
  // during which the workgroup holds a residency slot and feeds the matrix pipe nothing.  Priority outranks age: run the
  // prologue (and the epilogue) at raised priority, the k loop at the default.
  __builtin_amdgcn_s_setprio(SLIC_PRIO_EDGE);
  const int srow = tid / CPW;                                 // row inside each RALL-row group
  const int cq = (tid % CPW) ^ lds_swz<KD>(srow);             // SOURCE chunk column of this lane (LDS slot = tid % CPW)
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wgt = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, 0, p.wgt_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFF00u;
  unsigned aoff[AL], rmask[AL];
#pragma unroll
  for (int i = 0; i < AL; ++i) {
    const int64_t m = m0 + srow + RALL * i;
    aoff[i] = 0; rmask[i] = 0;
    if (m < p.M) {
      unsigned rr = (unsigned)m;
      const int gc = (int)(rr % (unsigned)p.Gc); rr /= (unsigned)p.Gc;
      const int gb = (int)(rr % (unsigned)p.Gb); rr /= (unsigned)p.Gb;
      const int ga = (int)(rr % (unsigned)p.Ga); rr /= (unsigned)p.Ga;
      const int a0 = ga * p.sa, b0 = gb * p.sb, c0 = gc * p.sc;
      aoff[i] = ((((rr * p.Ts + a0) * p.Hs + b0) * p.Ws + c0) * (unsigned)p.Cs + cq * 4) * 4u;
      unsigned mk = 0;
#pragma unroll
      for (int o = -3; o <= 3; ++o) {
        mk |= ((unsigned)(a0 + o) < (unsigned)p.Ts ? 1u : 0u) << (o + 3);
        mk |= ((unsigned)(b0 + o) < (unsigned)p.Hs ? 1u : 0u) << (7 + o + 3);
        mk |= ((unsigned)(c0 + o) < (unsigned)p.Ws ? 1u : 0u) << (14 + o + 3);
      }
      rmask[i] = mk;
    }
  }
  unsigned woff[BL];
#pragma unroll
  for (int i = 0; i < BL; ++i) {
    const int n = n0 + srow + RALL * i;
    woff[i] = n < p.N ? ((unsigned)n * (unsigned)p.ldw + cq * 4) * 4u : OOB;
  }
  // split-K: workgroup z of the grid's z dimension reduces k-tiles [kt0, nk) of the GEMM and writes raw accumulators to slab[z]
  // (conv_splitk_finish sums the slabs in z order and runs the epilogue); one split (slab == NULL) covers everything
  const int nk_all = p.nchunks / CPW;
  const int kt0 = slab ? bzi * kt_per_split : 0;
  const int nk = slab ? min(nk_all, kt0 + kt_per_split) : nk_all;     // END of this workgroup's k-tile range
  const int tiles_per_tap = p.Cs / KD;
  // k-tile -> (tap, channel base): a shift and a mask when the tap holds a power-of-two number of k-tiles (every R3D-18 layer)
  // instead of a wave-uniform division per k-tile per wave
  const bool tpt_pow2 = (tiles_per_tap & (tiles_per_tap - 1)) == 0;
  const int tpt_shift = 31 - __builtin_clz(tiles_per_tap);
  auto tap_of = [&](int ktc) { return tpt_pow2 ? (ktc >> tpt_shift) : (ktc / tiles_per_tap); };
  // per-tap records are read through the constant address space with a wave-uniform index: scalar loads, which
  // never touch the vmcnt queue the DMAs are counted on
  const __attribute__((address_space(4))) i32x4* tapc = (const __attribute__((address_space(4))) i32x4*)p.tab;
  // issue the (AL + BL) DMAs of k-tile kt into the LDS tile at float offset `toff`
  // A k-tile index past the end (kt >= nk) issues the same number of DMAs with every offset out of range: zeros land
  // in a ring stage nobody reads (or that a dummy stage multiplies as 0 * 0), so the main loop has no conditional
  // around its DMA or MFMA groups — the accumulators then stay in AGPRs for the whole loop.
  auto issue = [&](int kt, int toff) {
    const bool live = kt < nk;
    const int ktc = live ? kt : 0;
    const int tap = tap_of(ktc);                             // wave-uniform -> scalar loads of the tap record
    const int cb = (ktc - tap * tiles_per_tap) * KD;         // channel base inside the tap
    const i32x4 e = tapc[tap];                               // {src delta, tap mask, weight base, -}: s_load (lgkmcnt queue)
    const unsigned tm = live ? (unsigned)e.y : 0xFFFFFFFFu;  // an all-ones tap mask fails every row's bounds test
    const unsigned dlt = (unsigned)(e.x + cb) * 4u;
    const unsigned wc = (unsigned)(e.z + cb) * 4u;
    float* As = lds + toff;
    float* Bs = As + BM * KD;
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const unsigned off = ((rmask[i] & tm) == tm) ? aoff[i] + dlt : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void*)(As + ((RALL / 4) * wave + RALL * i) * KD),
                                               16, (int)off, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BL; ++i) {
      const unsigned off = (woff[i] == OOB || !live) ? OOB : woff[i] + wc;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (__attribute__((address_space(3))) void*)(Bs + ((RALL / 4) * wave + RALL * i) * KD),
                                               16, (int)off, 0, 0, 0);
    }
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;
  const int r = lane & 31, h = lane >> 5;
  // the next stage's DMAs (and their address math) are issued BETWEEN the MFMA groups of the current tile instead of
  // in front of them, so a wave's matrix pipe is not idle while it computes addresses (matters at 1-2 waves / SIMD)
  auto compute_tile = [&](int toff, int ktn, int toffn) {
    const float* Ab = lds + toff;
    const float* Bb = Ab + BM * KD;
    const bool live = ktn < nk;
    const int ktc = live ? ktn : 0;
    const int tap = tap_of(ktc);
    const int cb = (ktc - tap * tiles_per_tap) * KD;
    const i32x4 e = tapc[tap];
    const unsigned tm = live ? (unsigned)e.y : 0xFFFFFFFFu;
    const unsigned dlt = (unsigned)(e.x + cb) * 4u, wc = (unsigned)(e.z + cb) * 4u;
    float* Asn = lds + toffn;
    float* Bsn = Asn + BM * KD;
    f32x4 a[2][TM], b[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[0][i] = *(const f32x4*)&Ab[lds_off<KD>(wm * WTM + i * 32 + r, h)];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[0][j] = *(const f32x4*)&Bb[lds_off<KD>(wn * WTN + j * 32 + r, h)];
    constexpr int ILVQ = SLIC_ILVQ < NG ? SLIC_ILVQ : NG;
    constexpr int NPER = (AL + BL + ILVQ - 1) / ILVQ;   // DMAs per MFMA group: all issued within the first ILVQ groups
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const int cur = q & 1, nxt = cur ^ 1;
      if (q < NG - 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[nxt][i] = *(const f32x4*)&Ab[lds_off<KD>(wm * WTM + i * 32 + r, 2 * (q + 1) + h)];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[nxt][j] = *(const f32x4*)&Bb[lds_off<KD>(wn * WTN + j * 32 + r, 2 * (q + 1) + h)];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][t], b[cur][j][t], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int d = q * NPER; d < (q + 1) * NPER && d < AL + BL; ++d) {
        if (d < AL) {
          const unsigned off = ((rmask[d] & tm) == tm) ? aoff[d] + dlt : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void*)(Asn + ((RALL / 4) * wave + RALL * d) * KD),
                                                   16, (int)off, 0, 0, 0);
        } else {
          const unsigned off = (woff[d - AL] == OOB || !live) ? OOB : woff[d - AL] + wc;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (__attribute__((address_space(3))) void*)(Bsn + ((RALL / 4) * wave + RALL * (d - AL)) * KD),
                                                   16, (int)off, 0, 0, 0);
        }
      }
    }
  };
  // prologue: STAGES - 1 stages in flight
  const int ns = nk - kt0;                                   // number of stages' worth of work
#pragma unroll
  for (int t = 0; t < STAGES - 1; ++t) issue(kt0 + t, t * STAGE_FLOATS);
  SLIC_STAMP(stamp_wg, 2);                                   // prologue DMAs issued
#ifdef SLIC_STAMPS
  unsigned long long stamp_wait_v = 0, stamp_wait_b = 0, stamp_loop0 = 0;
#endif
  __builtin_amdgcn_s_setprio(0);
  constexpr int PER_STAGE = AL + BL;                         // DMA instructions per stage per wave, always exactly this many
  // Branch-free steady state: the trip count is rounded up to whole rings; stages past the end multiply zeros.
  for (int s0 = 0; s0 < ns; s0 += STAGES) {
#pragma unroll
    for (int sidx = 0; sidx < STAGES; ++sidx) {              // unrolled: ring stages are compile-time, so the compiler can
      const int sg = s0 + sidx;                               // see that the ds_reads and the in-flight DMAs never alias
      // stage sg has landed once only the DMAs of the STAGES - 2 younger in-flight stages are outstanding
#ifdef SLIC_STAMPS
      const unsigned long long st_a = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PER_STAGE) : "memory");
      const unsigned long long st_m = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();
      const unsigned long long st_b = __builtin_amdgcn_s_memtime();
      stamp_wait_v += st_m - st_a;
      stamp_wait_b += st_b - st_m;
      if (sg == 0) { SLIC_STAMP(stamp_wg, 3); stamp_loop0 = st_b; }     // first k-tile landed
#else
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PER_STAGE) : "memory");
      __builtin_amdgcn_s_barrier();        // every wave's part of stage sg is in LDS; the previous stage is free
#endif
      compute_tile(sidx * STAGE_FLOATS, kt0 + sg + STAGES - 1, ((sidx + STAGES - 1) % STAGES) * STAGE_FLOATS);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the trailing all-zero DMAs must land before LDS is reused
  __syncthreads();
  SLIC_STAMP(stamp_wg, 4);                                   // k loop done
#ifdef SLIC_STAMPS
  if (threadIdx.x == 0 && slic_stamps_buf) {                 // cycles wave 0 spent at the counted vmcnt wait / at the barrier / in the loop
    slic_stamps_buf[(size_t)stamp_wg * 8 + 6] = (stamp_wait_v << 32) | (stamp_wait_b & 0xFFFFFFFFull);
    slic_stamps_buf[(size_t)stamp_wg * 8 + 7] = __builtin_amdgcn_s_memtime() - stamp_loop0;
  }
#endif
  __builtin_amdgcn_s_setprio(SLIC_PRIO_EDGE);
  if (slab) {
    float* out = slab + (int64_t)bzi * slab_zstride;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WTN + j * 32 + r;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int64_t m = m0 + wm * WTM + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
          if (m < p.M && n < p.N) out[m * p.N + n] = acc[i][j][g];
        }
    }
    return;
  }
  conv_epilogue<BM, BN, WM, WN, TM, TN>(p, acc, lds, m0, n0, wm, wn, r, h, tid);
#ifdef SLIC_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // diagnostic build only: the stores have left the wave
  SLIC_STAMP(stamp_wg, 5);
#endif
}

template <int BM, int BN, int WM, int WN, int STAGES = 2, int KD = 32>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(conv_dma_waves(BM, BN, STAGES, KD))))
void conv_gemm_dma_kernel(const SlicConvArgs p, const int xcd_remap, float* __restrict__ slab, const int kt_per_split) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  conv_gemm_dma_body<BM, BN, WM, WN, STAGES, KD>(p, xcd_remap, slab, kt_per_split, lds, blockIdx.x, gridDim.x, blockIdx.y,
                                                      blockIdx.z, p.M * (int64_t)p.N);
}

// Tail-split launch: the row blocks [0, nfull_rb) run whole (one workgroup per tile, ordinary epilogue) and are dispatched
// first; the remaining row blocks — the part of the grid that would otherwise be a last, partly filled round of the chip's
// residency slots — are cut S ways along K and dispatched behind them, so the tail of the launch is many short workgroups
// instead of a few long ones.  Their raw accumulators go to slab[z][row - nfull_rb * BM][n]; conv_splitk_finish (rb0 = nfull_rb)
// adds the S pieces in z order and runs the epilogue for those rows only.  nfull_rb = 0 is plain split-K.
template <int BM, int BN, int WM, int WN, int STAGES = 2, int KD = 32>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(conv_dma_waves(BM, BN, STAGES, KD))))
void conv_gemm_dma_tail_kernel(const SlicConvArgs p, float* __restrict__ slab, const int kt_per_split, const int nfull_rb,
                               const int gxf, const int ny, const int S, const int64_t slab_zstride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int L = blockIdx.x;
  const int nfull = gxf * ny;                           // gxf = nfull_rb rounded up to a multiple of 8 (XCD order)
  if (L < nfull) {
    const int bx = L % gxf, by = L / gxf;
    int mb = (bx & 7) * (gxf >> 3) + (bx >> 3);
    if (mb >= nfull_rb) return;
    conv_gemm_dma_body<BM, BN, WM, WN, STAGES, KD>(p, 0, nullptr, 0, lds, mb, gxf, by, 0, 0);
    return;
  }
  const int Lt = L - nfull;
  const int z = Lt % S, t = Lt / S;
  const int by = t % ny, rb = nfull_rb + t / ny;
  // rows are addressed absolutely inside the body: shift the slab base so that row nfull_rb * BM is its row 0
  conv_gemm_dma_body<BM, BN, WM, WN, STAGES, KD>(p, 0, slab - (int64_t)nfull_rb * BM * p.N, kt_per_split, lds, rb, 0, by, z,
                                                      slab_zstride);
}

// up to SLIC_CONV_MULTI_MAX independent GEMMs (same N, same tile) in one grid: blockIdx.z picks the GEMM
#define SLIC_CONV_MULTI_MAX 8
struct SlicConvArgsPack {
  SlicConvArgs a[SLIC_CONV_MULTI_MAX];
  int gx[SLIC_CONV_MULTI_MAX];     // row blocks of each GEMM (a multiple of 8 when the XCD order is on)
};
template <int BM, int BN, int WM, int WN, int STAGES = 2, int KD = 32>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(conv_dma_waves(BM, BN, STAGES, KD))))
void conv_gemm_dma_multi_kernel(const SlicConvArgsPack pk, const int xcd_remap) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int c = blockIdx.z;
  const int gdx = pk.gx[c];
  if ((int)blockIdx.x >= gdx) return;
  conv_gemm_dma_body<BM, BN, WM, WN, STAGES, KD>(pk.a[c], xcd_remap, nullptr, 0, lds, blockIdx.x, gdx, blockIdx.y, 0, 0);
}

// second pass of a split-K launch: accumulators = sum over the S slabs in slab order, then the ordinary epilogue
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_splitk_finish(const SlicConvArgs p, const float* __restrict__ slab, const int S,
                                                          const int rb0, const int64_t zs) {
  __shared__ __attribute__((aligned(16))) float lds[conv_epi_lds_floats(BM, BN)];
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int64_t m0 = (int64_t)(rb0 + blockIdx.x) * BM;          // slab row 0 = row rb0 * BM of the GEMM
  const int n0 = blockIdx.y * BN;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WTN + j * 32 + r;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int64_t m = m0 + wm * WTM + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
        float a = 0.f;
        if (m < p.M && n < p.N) {
          const float* sp = slab + (m - (int64_t)rb0 * BM) * p.N + n;
          int z = 0;
          for (; z + 4 <= S; z += 4) {           // four loads in flight; the adds stay in slab order
            const float s0 = sp[z * zs], s1 = sp[(z + 1) * zs], s2 = sp[(z + 2) * zs], s3 = sp[(z + 3) * zs];
            a += s0; a += s1; a += s2; a += s3;
          }
          for (; z < S; ++z) a += sp[z * zs];
        }
        acc[i][j][g] = a;
      }
  }
  conv_epilogue<BM, BN, WM, WN, TM, TN>(p, acc, lds, m0, n0, wm, wn, r, h, tid);
}

// ------------------------------------------------------------------------------------------
// Weight gradient: dWp[n][kidx] = sum_m A[m][kidx] * dY[m][n]   (A = the forward's gathered rows).
// Workgroup = 128 kidx x 64 n (G = 2 sub-tiles of 64 kidx), waves 2 x 2, reduction over a slice of m in chunks of 32
// positions.  Both LDS tiles are [32 m][64] (channel-contiguous, as they sit in HBM); the MFMA operands are read with
// ds_read_b32 (lanes = 32 consecutive channels: conflict-free).  Split over m: slabs [slice][kidx][n], reduced (and unpacked
// into the reference [N][Cin][taps] layout) by conv_wgrad_reduce in slice order.
// Both operand tiles move HBM -> LDS by the DMA path (buffer_load ... lds): no staging
// registers, no ds_writes, a STAGES-deep ring with counted vmcnt waits.  A wave's DMA instruction covers 4 rows x 64
// floats (lane-linear 16-byte slots), which IS the [32 m][64] tile layout, so no swizzle is needed.
//   ILV: the next tile's DMAs are issued one at a time between the MFMA groups of the tile being computed.
//   RT:  per-row {source byte offset, in-bounds mask} records come from args.row_tab (slic_conv_row_table) instead of
//        being decoded from the row index — the per-tile VALU work drops from ~110 to ~30 instructions, which is
//        what bounded this kernel (measured: removing the address math alone gave +15 %).
template <int G, int STAGES, bool ILV, bool RT>
__global__ __launch_bounds__(256) void conv_wgrad_dma_kernel(const SlicConvArgs p, const float* __restrict__ dy,
                                                             int ldy, unsigned dy_bytes, float* __restrict__ slab,
                                                             int m_per_split, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  static_assert(!RT || STAGES == 2, "row-table path: the counted waits assume a 2-stage ring");
  constexpr int SUB = 32 * 64;                       // one [32 m][64] sub-tile
  constexpr int STAGE_FLOATS = (G + 1) * SUB;        // G sub-tiles of X and one of dY
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the DMAs' LDS bases (M0) then need no v_readfirstlane each
  const int wk = wave >> 1, wn = wave & 1;
  const int nx = (p.nchunks + 16 * G - 1) / (16 * G), ny = (p.N + 63) / 64;
  const int L = blockIdx.x;
  if (L >= nx * ny * nsplit) return;
  const int bx = L % nx, by = (L / nx) % ny, bz = L / (nx * ny);
  const int kc0 = bx * (16 * G);
  const int n0 = by * 64;
  const int mbeg = bz * m_per_split;
  const int mend = min(mbeg + m_per_split, (int)p.M);
  const int cq = tid & 15, srow = tid >> 4;
  int ex[G];
  unsigned tmk[G];                                   // tap mask of the chunk; all ones (never satisfied) for a padding chunk
  bool cv[G];
  int oa[G], ob[G], oc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int q = kc0 + g * 16 + cq;
    const int4 e = q < p.nchunks ? ((const int4*)p.tab)[q] : make_int4(0, -1, 0, 0);
    ex[g] = e.x * 4;
    cv[g] = e.y >= 0;
    tmk[g] = (unsigned)e.y;
    oa[g] = (e.w & 255) - 128; ob[g] = ((e.w >> 8) & 255) - 128; oc[g] = ((e.w >> 16) & 255) - 128;
  }
  const bool nvalid = (n0 + cq * 4) < p.N;
  const unsigned ycol = (unsigned)(n0 + cq * 4) * 4u;
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);
  // dY and the row table are range-checked against the END OF THIS SLICE: rows >= mend read as zeros / mask 0 with no
  // compare and no branch in the loop
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(
      (void*)dy, 0, (int)min((unsigned)dy_bytes, (unsigned)mend * (unsigned)(ldy * 4)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_rt = __builtin_amdgcn_make_buffer_rsrc((void*)p.row_tab, 0, RT ? mend * 8 : 0, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFF00u;
  // ---- row state: either decoded coordinates stepped in mixed radix, or records prefetched from the row table
  int rn[2], ra[2], rb[2], rc[2];
  int st_c = 0, st_b = 0, st_a = 0, st_n = 0;
  uint2 rec[2], recn[2];                             // RT: records of the next tile to issue / the one after
  auto load_rec = [&](int t, uint2 (&dst)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = mbeg + t * 32 + srow + 16 * i;            // past the slice: {0, 0}, and mask 0 fails every tap's test
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_rt, m * 8, 0, 0);
      dst[i] = make_uint2(v.x, v.y);
    }
  };
  if constexpr (RT) {
    load_rec(0, rec);
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned rr = (unsigned)(mbeg + srow + 16 * i);
      rc[i] = (int)(rr % (unsigned)p.Gc); rr /= (unsigned)p.Gc;
      rb[i] = (int)(rr % (unsigned)p.Gb); rr /= (unsigned)p.Gb;
      ra[i] = (int)(rr % (unsigned)p.Ga); rr /= (unsigned)p.Ga;
      rn[i] = (int)rr;
    }
    unsigned t = 32u;
    st_c = (int)(t % (unsigned)p.Gc); t /= (unsigned)p.Gc;
    st_b = (int)(t % (unsigned)p.Gb); t /= (unsigned)p.Gb;
    st_a = (int)(t % (unsigned)p.Ga); t /= (unsigned)p.Ga;
    st_n = (int)t;
  }
  // DMA `d` (of 2 * (G + 1)) of tile t into ring stage `stage`; tiles past the slice's end issue all-OOB DMAs (zeros)
  auto issue_piece = [&](int t, int stage, int d) {
    float* base = lds + stage * STAGE_FLOATS;
    const int i = d / (G + 1), g = d % (G + 1);
    const int m = mbeg + t * 32 + srow + 16 * i;
    const bool mv = m < mend;
    if (g < G) {
      unsigned off;
      if constexpr (RT) {
        off = ((rec[i].y & tmk[g]) == tmk[g]) ? rec[i].x + (unsigned)ex[g] : OOB;
      } else {
        const int a0 = ra[i] * p.sa, b0 = rb[i] * p.sb, c0 = rc[i] * p.sc;
        const unsigned rbase = ((((unsigned)rn[i] * p.Ts + a0) * p.Hs + b0) * p.Ws + c0) * (unsigned)p.Cs * 4u;
        const bool ok = mv && cv[g] && (unsigned)(a0 + oa[g]) < (unsigned)p.Ts &&
                        (unsigned)(b0 + ob[g]) < (unsigned)p.Hs && (unsigned)(c0 + oc[g]) < (unsigned)p.Ws;
        off = ok ? rbase + (unsigned)ex[g] : OOB;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void*)(base + g * SUB + (4 * wave + 16 * i) * 64),
                                               16, (int)off, 0, 0, 0);
    } else {
      const unsigned yo = nvalid ? (unsigned)m * (unsigned)(ldy * 4) + ycol : OOB;   // rows >= mend: out of rs_dy's range
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, (__attribute__((address_space(3))) void*)(base + G * SUB + (4 * wave + 16 * i) * 64),
                                               16, (int)yo, 0, 0, 0);
    }
  };
  auto step = [&]() {               // advance to the next tile's rows
    if constexpr (RT) {
      rec[0] = recn[0]; rec[1] = recn[1];
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        rc[i] += st_c; const int c1 = rc[i] >= p.Gc; rc[i] -= c1 ? p.Gc : 0;
        rb[i] += st_b + c1; const int c2 = rb[i] >= p.Gb; rb[i] -= c2 ? p.Gb : 0;
        ra[i] += st_a + c2; const int c3 = ra[i] >= p.Ga; ra[i] -= c3 ? p.Ga : 0;
        rn[i] += st_n + c3;
      }
    }
  };
  f32x16 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[g][v] = 0.f;
  const int r = lane & 31, h = lane >> 5;
  const int nmt = (mend > mbeg) ? (mend - mbeg + 31) / 32 : 0;
  constexpr int NP = 2 * (G + 1);                            // DMA instructions per stage per wave, always exactly this many
  constexpr int PER_STAGE = NP + (RT ? 2 : 0);               // + the two record loads, which sit on the same vmcnt queue
#pragma unroll
  for (int t = 0; t < STAGES - 1; ++t) {
    if constexpr (RT) load_rec(t + 1, recn);
#pragma unroll
    for (int d = 0; d < NP; ++d) issue_piece(t, t, d);
    step();
  }
  for (int s0 = 0; s0 < nmt; s0 += STAGES) {
#pragma unroll
    for (int sidx = 0; sidx < STAGES; ++sidx) {
      const int sg = s0 + sidx;
      const int tn = sg + STAGES - 1, stn = (sidx + STAGES - 1) % STAGES;      // tile / ring stage to fill
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PER_STAGE) : "memory");
      __builtin_amdgcn_s_barrier();
      if constexpr (RT) load_rec(tn + 1, recn);
      if constexpr (!ILV) {
#pragma unroll
        for (int d = 0; d < NP; ++d) issue_piece(tn, stn, d);
        step();
      }
      const float* Xb = lds + sidx * STAGE_FLOATS + wk * 32 + r + h * 64;
      const float* Yb = lds + sidx * STAGE_FLOATS + G * SUB + wn * 32 + r + h * 64;
      // operands of two reduction steps per group (rows 4q + h and 4q + 2 + h), register double-buffered
      float a[2][G][2], b[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        b[0][u] = Yb[(2 * u) * 64];
#pragma unroll
        for (int g = 0; g < G; ++g) a[0][g][u] = Xb[g * SUB + (2 * u) * 64];
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int cur = q & 1, nxt = cur ^ 1;
        if (q < 7) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            b[nxt][u] = Yb[(4 * (q + 1) + 2 * u) * 64];
#pragma unroll
            for (int g = 0; g < G; ++g) a[nxt][g][u] = Xb[g * SUB + (4 * (q + 1) + 2 * u) * 64];
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int g = 0; g < G; ++g)
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][g][u], b[cur][u], acc[g], 0, 0, 0);
        if constexpr (ILV) {
          constexpr int PER = (NP + SLIC_WG_ILVQ - 1) / SLIC_WG_ILVQ;   // pieces spread over the first SLIC_WG_ILVQ groups, step() in group 7
#pragma unroll
          for (int d = q * PER; d < (q + 1) * PER && d < NP; ++d) issue_piece(tn, stn, d);
          if (q == 7) step();
          // keep the operand fetch of group q + 1 in FRONT of group q's MFMAs (left alone, the scheduler sinks the ds_reads
          // next to their first use: no prefetch distance at all)
          if (q < 7) __builtin_amdgcn_sched_group_barrier(0x100, G + 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2 * G, 0);
        } else {
          if (q < 7) __builtin_amdgcn_sched_group_barrier(0x100, G + 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2 * G, 0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const int Kp = p.nchunks * 4;
  float* out = slab + (int64_t)bz * p.N * Kp;        // slab[z][kidx][n]
  const int n = n0 + wn * 32 + r;
  if (n < p.N) {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int kidx = (kc0 + g * 16) * 4 + wk * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
        if (kidx < Kp) out[(int64_t)kidx * p.N + n] = acc[g][v];
      }
  }
}


template <int G, int STAGES, bool ILV = false, bool RT = false>
static int launch_wgrad_dma(const SlicConvArgs& a, const float* dy, int ldy, unsigned dyb, float* slab, int per, int S,
                            hipStream_t st) {
  const size_t lds = (size_t)STAGES * (G + 1) * 32 * 64 * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<G, STAGES, ILV, RT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int64_t total = slic_cdiv(a.nchunks, 16 * G) * slic_cdiv(a.N, 64) * S;
  conv_wgrad_dma_kernel<G, STAGES, ILV, RT><<<dim3((unsigned)total), dim3(256), lds, st>>>(a, dy, ldy, dyb, slab, per, S);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

// row_tab[m] = {byte offset of row m's source origin, 21-bit in-bounds mask}  (see slic_conv_row_table)
__global__ void conv_row_table_kernel(const SlicConvArgs p, uint2* __restrict__ row_tab) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= p.M) return;
  unsigned rr = (unsigned)m;
  const int gc = (int)(rr % (unsigned)p.Gc); rr /= (unsigned)p.Gc;
  const int gb = (int)(rr % (unsigned)p.Gb); rr /= (unsigned)p.Gb;
  const int ga = (int)(rr % (unsigned)p.Ga); rr /= (unsigned)p.Ga;
  const int a0 = ga * p.sa, b0 = gb * p.sb, c0 = gc * p.sc;
  unsigned mk = 0;
#pragma unroll
  for (int o = -3; o <= 3; ++o) {
    mk |= ((unsigned)(a0 + o) < (unsigned)p.Ts ? 1u : 0u) << (o + 3);
    mk |= ((unsigned)(b0 + o) < (unsigned)p.Hs ? 1u : 0u) << (7 + o + 3);
    mk |= ((unsigned)(c0 + o) < (unsigned)p.Ws ? 1u : 0u) << (14 + o + 3);
  }
  row_tab[m] = make_uint2(((((rr * p.Ts + a0) * p.Hs + b0) * p.Ws + c0) * (unsigned)p.Cs) * 4u, mk);
}

// dW[n][c][tap] (reference layout, C = real channel count) = sum over splits of slab[s][k][n], splits added in slab order.
// A thread owns four consecutive n of one k: 16-byte loads from each of the S slabs (four slabs in flight), then four
// scattered 4-byte writes into the reference layout (one write per element against S reads).
// K is cut into runs of RL floats holding PPR pixels (taps) x Cs channels: the ordinary operand has RL = Cs, PPR = 1
// (k = tap * Cs + c); the W-run operand of the RGB stem RL = 24, PPR = 7, Cs = 3.
template <bool RUNS>
__global__ __launch_bounds__(256) void conv_wgrad_reduce(const float* __restrict__ slab, int S, int N, int Kp, int Cs, int C,
                                                         int ntaps, int RL, int PPR, float* __restrict__ dW) {
  const int64_t e4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int N4 = N >> 2;
  const int64_t tot4 = (int64_t)Kp * N4;
  if (e4 >= tot4) return;
  const int n0 = (int)(e4 % N4) * 4;
  const int k = (int)(e4 / N4);
  int tap, c;
  if (RUNS) {
    const int run = k / RL, rem = k - run * RL;
    const int px = rem / Cs;
    c = rem - px * Cs;
    tap = run * PPR + px;
    if (px >= PPR) return;
  } else {
    tap = k / Cs;
    c = k - tap * Cs;
  }
  if (tap >= ntaps || c >= C) return;
  const f32x4* sp = (const f32x4*)(slab + (int64_t)k * N + n0);
  const int64_t zs = ((int64_t)N * Kp) >> 2;     // slab stride in f32x4 units
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  int z = 0;
  for (; z + 4 <= S; z += 4) {
    const f32x4 v0 = sp[z * zs], v1 = sp[(z + 1) * zs], v2 = sp[(z + 2) * zs], v3 = sp[(z + 3) * zs];
    a += v0; a += v1; a += v2; a += v3;
  }
  for (; z < S; ++z) a += sp[z * zs];
  float* d = dW + ((int64_t)n0 * C + c) * ntaps + tap;
  const int64_t ns = (int64_t)C * ntaps;
  d[0] = a.x; d[ns] = a.y; d[2 * ns] = a.z; d[3 * ns] = a.w;
}

// ---- weight packing -------------------------------------------------------------------------
// Both packers re-order the reference layout W[n][c][tap] through an LDS tile so that global reads are contiguous runs of W and
// global writes contiguous runs of the operand; the padding of the operands (channels C..Cs, columns past the last tap) is
// written once, as zeros, when the plan allocates them — the packers only touch real elements.
// forward:  Wp[n][tap*Cs + c] = W[n][c][tap].  Workgroup = (n, chunk of 64 channels); NT = compile-time tap count (0: generic),
// so the index arithmetic divides by constants.
template <int NT>
__global__ __launch_bounds__(256) void pack_w_fwd(const float* __restrict__ W, int N, int C, int ntaps_rt, int Cs, int Kp,
                                                  int CH, float* __restrict__ Wp) {
  extern __shared__ float pk_lds[];                        // [CH][ld], ld odd: conflict-free column reads
  const int ntaps = NT ? NT : ntaps_rt;
  const int n = blockIdx.x, c0 = blockIdx.y * CH, t = threadIdx.x;
  const int ld = ntaps | 1;
  const int cw = min(CH, C - c0);                          // real channels in this chunk
  if (cw <= 0) return;
  const float* src = W + ((int64_t)n * C + c0) * ntaps;
  for (int e = t; e < cw * ntaps; e += 256) {
    const int cc = e / ntaps, tap = e - cc * ntaps;
    pk_lds[cc * ld + tap] = src[e];
  }
  __syncthreads();
  float* dst = Wp + (int64_t)n * Kp + c0;
  if (cw == 64) {
    for (int e = t; e < ntaps * 64; e += 256) {
      const int tap = e >> 6, cc = e & 63;
      dst[tap * Cs + cc] = pk_lds[cc * ld + tap];
    }
  } else {
    for (int e = t; e < ntaps * cw; e += 256) {
      const int tap = e / cw, cc = e - tap * cw;
      dst[tap * Cs + cc] = pk_lds[cc * ld + tap];
    }
  }
}
// data gradient:  Wd[c][tap*N + n] = W[n][c][tap]: with r = c*ntaps + tap this is the 2-D transpose [N][R] -> [R][N] of W seen
// as an N x R matrix (R = C*ntaps), rows of the result laid out with stride N inside a Wd row of Kd floats.  64 x 64 tiles,
// 16-byte global accesses both ways (R % 4 == 0, N % 4 == 0), one division per output ROW.
__global__ __launch_bounds__(256) void pack_w_dgrad(const float* __restrict__ W, int N, int R, int ntaps, int Kd,
                                                    float* __restrict__ Wd) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.x * 64, n0 = blockIdx.y * 64, t = threadIdx.x;
  const int q = (t & 15) * 4, rowt = t >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int nn = rowt + 16 * i;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n0 + nn < N && r0 + q < R) v = *(const f32x4*)(W + (int64_t)(n0 + nn) * R + r0 + q);
    tile[nn][q] = v.x; tile[nn][q + 1] = v.y; tile[nn][q + 2] = v.z; tile[nn][q + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rr = rowt + 16 * i;
    const int r = r0 + rr;
    if (r < R && n0 + q < N) {
      const int c = r / ntaps, tap = r - c * ntaps;
      const f32x4 v = {tile[q][rr], tile[q + 1][rr], tile[q + 2][rr], tile[q + 3][rr]};
      *(f32x4*)(Wd + (int64_t)c * Kd + (int64_t)tap * N + n0 + q) = v;
    }
  }
}
// the same for shapes whose rows are not 16-byte multiples (C % 4 != 0: never the case for a layer that needs a data gradient in
// R3D-18 — its only few-channel layer is the stem): one thread per element
__global__ void pack_w_dgrad_any(const float* __restrict__ W, int N, int C, int ntaps, int Kd, float* __restrict__ Wd) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // (c, tap, n), n fastest
  if (e >= (int64_t)C * ntaps * N) return;
  const int n = (int)(e % N);
  const int64_t ct = e / N;
  const int tap = (int)(ct % ntaps), c = (int)(ct / ntaps);
  Wd[(int64_t)c * Kd + (int64_t)tap * N + n] = W[((int64_t)n * C + c) * ntaps + tap];
}
// W-run operand (few-channel stem): Wp[n][run * RL + px * C + c] = W[n][c][run * PPR + px]; a few hundred KB, one thread per element
__global__ void pack_w_fwd_runs(const float* __restrict__ W, int N, int C, int ntaps, int RL, int PPR, int Kp, float* __restrict__ Wp) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)N * Kp) return;
  const int k = (int)(e % Kp), n = (int)(e / Kp);
  const int run = k / RL, rem = k - run * RL;
  const int px = rem / C, c = rem - px * C;
  const int tap = run * PPR + px;
  Wp[e] = (px < PPR && tap < ntaps) ? W[((int64_t)n * C + c) * ntaps + tap] : 0.f;
}
// data gradient:  Wd[c][tap*N + n] = W[n][c][tap]   (rows = input channels, Cs rows, zero padded).
// Workgroup = (32 output channels n, CB input channels c): reads CB*ntaps contiguous floats per n, writes 32 consecutive n.
__global__ __launch_bounds__(256) void pack_w_dgrad(const float* __restrict__ W, int N, int C, int ntaps, int Cs, int Kd,
                                                    int CB, float* __restrict__ Wd) {
  extern __shared__ float pk_lds[];                        // [32][ld]
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * CB, t = threadIdx.x;
  const int run = CB * ntaps;
  const int ld = run | 1;
  const int cw = min(CB, C - c0);                          // real channels (<= 0: padding rows)
  const int cbw = min(CB, Cs - c0);
  const int nw = min(32, N - n0);
  const int rw = cw > 0 ? cw * ntaps : 0;
  for (int e = t; e < nw * rw; e += 256) {
    const int nn = e / rw, r = e - nn * rw;
    pk_lds[nn * ld + r] = W[((int64_t)(n0 + nn) * C + c0) * ntaps + r];
  }
  __syncthreads();
  for (int e = t; e < cbw * ntaps * 32; e += 256) {
    const int nn = e & 31, q = e >> 5;                     // q = cc * ntaps + tap
    const int cc = q / ntaps, tap = q - cc * ntaps;
    if (nn < nw) Wd[(int64_t)(c0 + cc) * Kd + tap * N + n0 + nn] = cc < cw ? pk_lds[nn * ld + q] : 0.f;
  }
  if (blockIdx.x == 0)
    for (int cc = 0; cc < cbw; ++cc)
      for (int k = ntaps * N + t; k < Kd; k += 256) Wd[(int64_t)(c0 + cc) * Kd + k] = 0.f;
}

// ---- layout conversion ------------------------------------------------------------------------
// NCDHW [B, C, S] -> NDHWC [B, S, Cp] (S = T*H*W, channels zero-padded to Cp)
__global__ void ncdhw_to_ndhwc(const float* __restrict__ x, int B, int C, int64_t S, int Cp,
                               float* __restrict__ y) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)B * S) return;
  const int64_t b = e / S, s = e % S;
  for (int c = 0; c < Cp; ++c)
    y[e * Cp + c] = c < C ? x[(b * C + c) * S + s] : 0.f;
}

// NCDHW [B, C, R, W] -> [B, R, Wp, C]: rows of (T*H), `pad` zero columns on the left, zeros up to Wp on the right (the W-run stem operand)
__global__ void ncdhw_to_ndhwc_wpad(const float* __restrict__ x, int B, int C, int64_t R, int W, int pad, int Wp, float* __restrict__ y) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // (b, row, wp)
  if (e >= (int64_t)B * R * Wp) return;
  const int wp = (int)(e % Wp);
  const int64_t br = e / Wp;
  const int64_t b = br / R, row = br % R;
  const int w = wp - pad;
  const bool in = w >= 0 && w < W;
  for (int c = 0; c < C; ++c) y[e * C + c] = in ? x[((b * C + c) * R + row) * W + w] : 0.f;
}

// ------------------------------------ C ABI ------------------------------------------------
static inline hipStream_t S_(void* s) { return (hipStream_t)s; }

// ------------------------------------------------------------------------------------------
// Winograd F(4, 3) along W for the 3 x 3 x 3, stride-1, pad-1 convolutions (layer1 / layer2: W = 56 / 28) — forward and,
// with the flipped / transposed operand, data gradient.  Exact fp32 arithmetic (v_mfma_f32_32x32x2_f32), half the multiplies:
// four outputs along W and the three kw taps cost six products per (kt, kh, c, n) instead of twelve.
//   A GEMM row is a W-TILE (b, t, h, wt): outputs w = 4 wt .. 4 wt + 3.  For every (kt, kh) and channel c the six input pixels
//   d[a] = x[t + kt - 1, h + kh - 1, 4 wt - 1 + a, c] are transformed in registers, V = B^T d (six points), each point p is its
//   own GEMM  M_p[tile][n] += V_p[tile][(kt, kh, c)] * U_p[(kt, kh, c)][n]  with  U_p = sum_kw G[p][kw] w[n][c][kt][kh][kw]
//   (packed once per step by pack_w_wino), six accumulators per wave tile, and the outputs are Y = A^T M in registers.
//   Workgroup: 64 W-tiles x 64 n, 2 x 2 waves of 32 x 32 (6 x 16 accumulator registers); K loop = 9 (kt, kh) x C / 8 stages of
//   8 channels; a stage is 12 KB of raw pixels [a 6][h 2][tile 64][4 ch] + 12 KB of U [p 6][h 2][n 64][4 ch], both by LDS-DMA in
//   the order the lanes read them (lane (r, h) of the MFMA fetches channels 4 h .. 4 h + 3 of tile / column r with ONE
//   ds_read_b128 per pixel / point and feeds element j to MFMA j: k = 0 / 1 of MFMA j are channels j / 4 + j); 3-stage ring,
//   counted vmcnt, one barrier per stage, the last two points of a stage multiplied after the NEXT stage's barrier; the transform is
//   24 packed-fp32 instructions per stage (wino_bt6), the DMA addressing three vector instructions per pixel piece and none per U
//   piece — a wave's VALU instructions do not hide under its own MFMAs on this part.
//   Epilogue: the 256 x 64 outputs leave through the shared LDS-image epilogue in two 128-row halves (conv_epilogue_rows: store /
//   addend / mask / BatchNorm partials with slab rows of 128 GEMM rows, as variant 22).
// ------------------------------------------------------------------------------------------
#include "wino_common.h"

#ifndef SLIC_WINO_ABL
#define SLIC_WINO_ABL 0   // diagnostic builds only (csrc/_exp/, scripts/r3/ab_wino.sh): 1 = DMAs out of range, 2 = no stage barrier, 4 = cache-resident input,
                          // 8 = pre-transformed operand emulation (no transform, no validity VALU), 16 = with 8: record-contiguous operand layout
#endif

// TG = groups of 32 W-tiles per workgroup (two waves each, one per n half): 2 (256 threads, 2 workgroups / CU) or 4 (512 threads, ONE
// workgroup / CU, the same two waves per SIMD: the 12 KB U stage — identical for every workgroup of a launch — is then fetched once
// per 128 tiles instead of once per 64, and the ring holds the same bytes per CU)
constexpr int wino_stage_floats(int TG) { return 12 * 32 * TG * 4 + (TG == 4 ? 1024 : 768) * 4; }

template <int STAGES, bool WPAD, int TG>
__global__ __launch_bounds__(128 * TG) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_wino_kernel(const SlicConvArgs p, float* __restrict__ slab, const int st_per_split, const int mb_off, const int mb_cnt) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the DMAs' LDS bases (M0) then need no v_readfirstlane
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int bx = blockIdx.x, gdx = gridDim.x;
  const int mb = (bx & 7) * (gdx >> 3) + (bx >> 3);          // XCD-aware order (see conv_gemm_dma_body)
  const int Wq = (p.Ws + 3) >> 2;                             // W-tiles per row (WPAD: the last one is ragged)
  const int64_t Mt = WPAD ? (p.M / p.Ws) * Wq : (p.M >> 2);   // W-tiles
  constexpr int TW = 32 * TG;                                 // W-tiles per workgroup
  constexpr int NT = 128 * TG;                                // threads
  constexpr int A_FLOATS = 12 * TW * 4;
  constexpr int STAGE_FLOATS = wino_stage_floats(TG);
  constexpr int UP = TG == 4 ? 2 : 3;                         // U pieces per thread (TG = 4: 1024 chunk slots for 768 chunks)
  constexpr int NPC = 3 + UP;                                 // DMA pieces per thread and stage
  // this launch covers tile blocks [mb_off, mb_off + mb_cnt): all of them, or — a launch whose last dispatch round would be partly
  // filled — the whole rounds (K loop in one piece) and then the remaining blocks with the K loop cut (launch_wino)
  if (mb >= mb_cnt) return;
  const int64_t tile0 = (int64_t)(mb_off + mb) * TW;
  if (tile0 >= Mt) return;
  const int nb = blockIdx.y, n0 = nb * 64;
  __builtin_amdgcn_s_setprio(SLIC_PRIO_EDGE);
  const int C = p.Cs, T = p.Ts, H = p.Hs, W = p.Ws;
  const int CCH = C >> 3;                                     // 8-channel stages per (kt, kh)
  // split-K (few-tile launches: layer4): workgroup z of the grid's z dimension reduces stages [sbeg, NS) of the K loop, transforms
  // its partial accumulators (A^T is linear) and writes the outputs to slab[z][GEMM row][n]; conv_splitk_finish adds the pieces in z
  // order and runs the epilogue.  One piece (slab == NULL) covers everything.
  const int NS_all = 9 * CCH;
  const int sbeg = slab ? (int)blockIdx.z * st_per_split : 0;
  const int NS = slab ? min(NS_all, sbeg + st_per_split) : NS_all;      // END of this workgroup's stage range
  const int NB = p.N >> 6;
  // ---- DMA roles: chunk q = i * NT + tid of the pixel image [a 6][slot 2 TW][4 ch]: thread = (slot tid % (2 TW), a = 2 i + tid / (2 TW)).
  // Slot 2 tile + (half ^ bit 3 of the tile): the two 16-byte halves of a pixel's 8 channels are fetched by NEIGHBOURING lanes (one
  // 32-byte piece of a cache line instead of two lines' worth of lookups in two instructions), and the xor keeps the readers'
  // ds_read_b128 conflict-free (lanes r and r + 8 of a 16-lane phase would otherwise share banks at the 32-byte tile stride)
  const int myslot = tid % (2 * TW), mytl = myslot >> 1, myhc = (myslot & 1) ^ ((mytl >> 3) & 1);
#if SLIC_WINO_ABL & 4
  const int64_t mytile = (tile0 & 0x3C0) + mytl;     // diagnostic build: every workgroup reads one of 16 tile blocks (cache-resident input)
#else
  const int64_t mytile = tile0 + mytl;
#endif
  const bool tvalid = mytile < Mt;
  unsigned q = (unsigned)(tvalid ? mytile : 0);
  const int wt = (int)(q % (unsigned)Wq); q /= (unsigned)Wq;
  const int hh = (int)(q % (unsigned)H); q /= (unsigned)H;
  const int tt = (int)(q % (unsigned)T); q /= (unsigned)T;    // q = batch
  unsigned aoff[3];
  bool avalid[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int a = 2 * i + tid / (2 * TW), hc = myhc;
    const int w = 4 * wt - 1 + a;
    avalid[i] = tvalid && (unsigned)w < (unsigned)W;
    aoff[i] = (unsigned)((((((int64_t)q * T + tt) * H + hh) * W + w) * C + 4 * hc) * 4);
  }
  unsigned tmask = 0, hmask = 0;
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    tmask |= ((unsigned)(tt + o - 1) < (unsigned)T ? 1u : 0u) << o;
    hmask |= ((unsigned)(hh + o - 1) < (unsigned)H ? 1u : 0u) << o;
  }
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wgt = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, 0, p.wgt_bytes, 0x00020000);
  [[maybe_unused]] constexpr unsigned OOB = 0xFFFFFF00u;
  const int cch_shift = 31 - __builtin_clz(CCH);             // CCH is a power of two (checked on the host)
  const unsigned uoff0 = (unsigned)tid * 16u;                 // this lane's 16 bytes of a U stage block, piece 0
  // Validity of pixel piece i at (kt, kh), as the bit tap9 = 3 kt + kh of an INVALID mask: v_bfe_i32 of one bit gives 0 / -1, and
  // offset | -1 is out of range.  Bit 9 (the tap of a dead stage) is set.
  unsigned inv9[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    unsigned m = 1u << 9;
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      const bool ok = avalid[i] && ((tmask >> (t9 / 3)) & 1u) && ((hmask >> (t9 % 3)) & 1u);
      m |= (ok ? 0u : 1u) << t9;
    }
    inv9[i] = m;
  }
  unsigned uvoff[UP];
#pragma unroll
  for (int i = 0; i < UP; ++i) uvoff[i] = (i * NT + tid) < 768 ? uoff0 + (unsigned)(i * NT * 16) : 0xFFFFFF00u;    // past the block: zeros
  // The six DMA pieces of stage s into the ring slot at float offset `toff`: pieces 0-2 raw pixels, 3-5 U.  Per stage everything
  // but the validity test is scalar: pixels = (lane base + scalar delta) | (invalid ? -1 : 0) — three vector instructions per piece;
  // U = a per-lane constant offset + the stage's block as the instruction's SCALAR offset — none.  A dead stage (s >= NS) reads
  // tap 9: every pixel piece out of range (zeros), U block 0 (finite values times zeros).
  struct StageRec { unsigned delta, ublk; int tap; };
  auto stage_rec = [&](int s) {
    StageRec q;
    const bool live = s < NS;
    const int sc = live ? s : 0;
    const int tap9 = sc >> cch_shift, cc = sc & (CCH - 1);
    const int kt = (tap9 * 11) >> 5, kh = tap9 - 3 * kt;
    q.tap = live ? tap9 : 9;
    q.delta = (unsigned)((((kt - 1) * H + (kh - 1)) * W * C + cc * 8) * 4);
    q.ublk = (unsigned)((tap9 * CCH + cc) * NB + nb) * (unsigned)(12 * 64 * 4 * 4);
    return q;
  };
  auto issue_piece = [&](const StageRec& q, int toff, int d) {
    if (d < 3) {
#if SLIC_WINO_ABL & 8
      // diagnostic build (wrong results, right timing): what a kernel reading a PRE-TRANSFORMED, zero-padded operand would issue —
      // a per-lane constant offset + the stage's offset in the instruction's scalar operand, no vector instruction per piece.
      // & 16: the operand laid out [row (b, t, h)][C / 8][W-tile][6 points][8 ch] (a stage's 64 tiles are whole 192-byte records,
      // contiguous along a row) instead of today's 32-byte pieces at channel stride
#if SLIC_WINO_ABL & 16
      const unsigned qd = (unsigned)(d * NT + tid), tl_ = qd / 12u, within = qd % 12u;
      const unsigned rowrel = (unsigned)(((tile0 % Wq) + tl_) / (unsigned)Wq), wt_ = (unsigned)(((tile0 % Wq) + tl_) % (unsigned)Wq);
      const unsigned voff = ((rowrel * (unsigned)CCH) * (unsigned)Wq + wt_) * 192u + within * 16u;
      const unsigned row0 = (unsigned)((tile0 / Wq) * 2 / 3);
      const int tap9_ = q.tap == 9 ? 0 : q.tap;
      const int kt_ = (tap9_ * 11) >> 5, kh_ = tap9_ - 3 * kt_;
      const unsigned cc_ = (q.delta >> 5) & (unsigned)(CCH - 1);
      const unsigned soff = (q.tap == 9) ? 0xFFFFFF00u : ((row0 + (unsigned)(kt_ * H + kh_)) * (unsigned)CCH + cc_) * (unsigned)Wq * 192u;
#else
      const unsigned voff = avalid[d] ? aoff[d] : 0u;
      const unsigned soff = (q.tap == 9) ? 0xFFFFFF00u : q.delta + (unsigned)((H * W + W) * C * 4);
#endif
#if SLIC_WINO_ABL & 1
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void*)(lds + toff + (d * NT + wave * 64) * 4),
                                               16, (int)(OOB + 0 * voff), (int)(0 * soff), 0, 0);
#else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void*)(lds + toff + (d * NT + wave * 64) * 4),
                                               16, (int)voff, (int)soff, 0, 0);
#endif
#elif SLIC_WINO_ABL & 1
      const unsigned off = OOB + 0 * (aoff[d] + q.delta);     // diagnostic build: DMAs issued, no memory traffic
#else
      const unsigned off = (aoff[d] + q.delta) | (unsigned)__builtin_amdgcn_sbfe(inv9[d], q.tap, 1);
#endif
#if !(SLIC_WINO_ABL & 8)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void*)(lds + toff + (d * NT + wave * 64) * 4),
                                               16, (int)off, 0, 0, 0);
#endif
    } else {
      const int i = d - 3;
#if SLIC_WINO_ABL & 1
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (__attribute__((address_space(3))) void*)(lds + toff + A_FLOATS + (i * NT + wave * 64) * 4),
                                               16, (int)(OOB + 0 * q.ublk), 0, 0, 0);
#else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (__attribute__((address_space(3))) void*)(lds + toff + A_FLOATS + (i * NT + wave * 64) * 4),
                                               16, (int)uvoff[i], (int)q.ublk, 0, 0);
#endif
    }
  };
  f32x16 acc[6];
#pragma unroll
  for (int pp = 0; pp < 6; ++pp)
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[pp][g] = 0.f;
#pragma unroll
  for (int t = 0; t < STAGES - 1; ++t) {
    const StageRec q = stage_rec(sbeg + t);
#pragma unroll
    for (int d = 0; d < NPC; ++d) issue_piece(q, t * STAGE_FLOATS, d);
  }
  __builtin_amdgcn_s_setprio(0);
  constexpr int AP = 2 * TW * 4;                              // floats between pixels a and a + 1 of the image
  const int aro = (2 * (wm * 32 + r) + (h ^ ((r >> 3) & 1))) * 4;   // + a * AP: pixel a of this lane's tile, its channel half
  const int bro = A_FLOATS + (h * 64 + wn * 32 + r) * 4;      // + p * 512: point p of this lane's column, its channel half
  // Software pipeline across the stage barrier: the last two points of stage s - 1 (operands already in registers) are multiplied
  // AFTER the barrier of stage s, under the latency of stage s's first LDS reads — the barrier then sits where no wave needs
  // anything from LDS for the next 8 MFMAs, and a slot is free for the DMAs of stage s + STAGES - 1 as soon as the barrier is
  // passed (every wave's LDS reads of stage s - 1 were issued before it).
  f32x4 bt0 = {0.f, 0.f, 0.f, 0.f}, bt1 = bt0;               // U of points 4, 5 of the previous stage
  const f32x2 c2 = {2.f, 2.f}, c4 = {4.f, 4.f}, c5 = {5.f, 5.f};
  // V of the previous stage (channel pairs (0, 1) and (2, 3) of the lane's four): points 4, 5 are multiplied after the barrier
  f32x2 Vl[6], Vh[6];
#pragma unroll
  for (int pp = 0; pp < 6; ++pp) { Vl[pp] = (f32x2){0.f, 0.f}; Vh[pp] = (f32x2){0.f, 0.f}; }
  // four MFMAs each of points p0, p1: element j of the lane's four channels goes to MFMA j
  auto mfma_pair = [&](const int p0, const int p1, const f32x4& u0, const f32x4& u1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[p0] = __builtin_amdgcn_mfma_f32_32x32x2f32(j < 2 ? Vl[p0][j] : Vh[p0][j - 2], u0[j], acc[p0], 0, 0, 0);
      acc[p1] = __builtin_amdgcn_mfma_f32_32x32x2f32(j < 2 ? Vl[p1][j] : Vh[p1][j - 2], u1[j], acc[p1], 0, 0, 0);
    }
  };
  for (int s0 = sbeg; s0 < NS; s0 += STAGES) {
#pragma unroll
    for (int sidx = 0; sidx < STAGES; ++sidx) {
      const int sg = s0 + sidx;
      // stage sg has landed; and this wave's LDS reads of stage sg - 1 are COMPLETE (lgkmcnt(0)), not merely issued, before the
      // barrier lets another wave's DMAs overwrite that slot
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((STAGES - 2) * NPC) : "memory");
#if !(SLIC_WINO_ABL & 2)
      __builtin_amdgcn_s_barrier();
#endif
      const float* St = lds + sidx * STAGE_FLOATS;
      const int toffn = ((sidx + STAGES - 1) % STAGES) * STAGE_FLOATS;
      const StageRec qn = stage_rec(sg + STAGES - 1);
      const f32x4 d0 = *(const f32x4*)&St[aro], d1 = *(const f32x4*)&St[aro + AP], d2 = *(const f32x4*)&St[aro + 2 * AP];
      const f32x4 d3 = *(const f32x4*)&St[aro + 3 * AP], d4 = *(const f32x4*)&St[aro + 4 * AP], d5 = *(const f32x4*)&St[aro + 5 * AP];
      const f32x4 b0 = *(const f32x4*)&St[bro], b1 = *(const f32x4*)&St[bro + 512];
      mfma_pair(4, 5, bt0, bt1);                               // the previous stage's last two points, under the latency of these reads
      // V = B^T d in packed pairs (wino_bt6).  The packed ops are inline assembly, which the compiler's hazard recogniser does not
      // see as VALU: an MFMA reading a VGPR within two instructions of the VALU instruction that wrote it reads the OLD value (the
      // compiler puts `s_nop 1` there for its own instructions).  So the transform is one block fenced off from the MFMAs that
      // consume it, closed by the two wait states.
      const f32x2 dl[6] = {{d0[0], d0[1]}, {d1[0], d1[1]}, {d2[0], d2[1]}, {d3[0], d3[1]}, {d4[0], d4[1]}, {d5[0], d5[1]}};
      const f32x2 dh[6] = {{d0[2], d0[3]}, {d1[2], d1[3]}, {d2[2], d2[3]}, {d3[2], d3[3]}, {d4[2], d4[3]}, {d5[2], d5[3]}};
#if SLIC_WINO_ABL & 8
#pragma unroll
      for (int pp = 0; pp < 6; ++pp) { Vl[pp] = dl[pp]; Vh[pp] = dh[pp]; }
#else
      __builtin_amdgcn_sched_barrier(0);
      wino_bt6(dl, Vl, c2, c4, c5);
      wino_bt6(dh, Vh, c2, c4, c5);
      asm volatile("s_nop 1" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#endif
      const f32x4 b2 = *(const f32x4*)&St[bro + 2 * 512], b3 = *(const f32x4*)&St[bro + 3 * 512];
      mfma_pair(0, 1, b0, b1);
#pragma unroll
      for (int d = 0; d < 3; ++d) issue_piece(qn, toffn, d);
      bt0 = *(const f32x4*)&St[bro + 4 * 512];
      bt1 = *(const f32x4*)&St[bro + 5 * 512];
      mfma_pair(2, 3, b2, b3);
#pragma unroll
      for (int d = 3; d < NPC; ++d) issue_piece(qn, toffn, d);
    }
  }
  mfma_pair(4, 5, bt0, bt1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __builtin_amdgcn_s_setprio(SLIC_PRIO_EDGE);
  // Y = A^T M,  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]   (in place: acc[0..3] become the four outputs)
  {
    const f32x16 s12 = acc[1] + acc[2], d12 = acc[1] - acc[2], s34 = acc[3] + acc[4], d34 = acc[3] - acc[4];
    acc[0] = acc[0] + s12 + s34;
    acc[1] = d12 + 2.f * d34;
    acc[2] = s12 + 4.f * s34;
    acc[3] = d12 + 8.f * d34 + acc[5];
  }
  if (slab) {
    // slab[piece][GEMM row - first row of this launch][n]  (a tail launch exists on widths that are multiples of 4 only: 4 rows per tile)
    const int64_t row_off = (int64_t)mb_off * TW * 4;
    float* out = slab + ((int64_t)blockIdx.z * (p.M - row_off) - row_off) * p.N;
    const int n = n0 + wn * 32 + r;
    const int64_t bth_all = p.M / W;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int64_t tile = tile0 + wm * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
      const int64_t bth = tile / Wq;
      const int wq4 = (int)(tile - bth * Wq) * 4;
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (bth < bth_all && wq4 + o < W) out[(bth * W + wq4 + o) * p.N + n] = acc[o][g];
    }
    return;
  }
#pragma unroll
  for (int hf = 0; hf < TG; ++hf) {
    // a 128-row block past the last W-tile does not exist: it has no slab row in stat_partial / bwd_partial (workgroup-uniform)
    if (tile0 + hf * 32 >= Mt) break;
    if (wm == hf) {
      const int col = wn * 32 + r;
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int trow = (g & 3) + 8 * (g >> 2) + 4 * h;
          lds[(4 * trow + o) * 64 + col] = acc[o][g];
        }
    }
    __syncthreads();
    conv_epilogue_rows<128, 64, WPAD, 128 * TG>(p, lds, (tile0 + hf * 32) * 4, n0, tid);
    __syncthreads();
  }
}

// U[(((tap9 * C/8 + cc) * N/64 + nb) * 12 + p * 2 + h) * 64 + nl][j] = sum_kw G[p][kw] * w(n = 64 nb + nl, c = 8 cc + 4 h + j, kt, kh, kw)
//   forward : w(n, c, kt, kh, kw) = W[n][c][kt][kh][kw]                    (N_ = out channels N, C_ = in channels C)
//   dgrad   : w(n, c, kt, kh, kw) = W[c][n][2 - kt][2 - kh][2 - kw]        (N_ = C: channels of dx, C_ = N: channels of dy)
// G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
__global__ void pack_w_wino(const float* __restrict__ Wt, int N, int C, int dgrad, float* __restrict__ U) {
  const int N_ = dgrad ? C : N, C_ = dgrad ? N : C;
  const int64_t tot = (int64_t)9 * C_ * N_;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= tot) return;
  // e = ((((tap9 * CCH + cc) * NB + nb) * 2 + h) * 64 + nl) * 4 + j
  int64_t q = e;
  const int j = (int)(q & 3); q >>= 2;
  const int nl = (int)(q & 63); q >>= 6;
  const int h = (int)(q & 1); q >>= 1;
  const int NB = N_ >> 6, CCH = C_ >> 3;
  const int nb = (int)(q % NB); q /= NB;
  const int cc = (int)(q % CCH); q /= CCH;
  const int tap9 = (int)q;
  const int kt = tap9 / 3, kh = tap9 % 3;
  const int n = nb * 64 + nl, c = cc * 8 + 4 * h + j;
  float w[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    if (dgrad) w[kw] = Wt[((int64_t)c * C + n) * 27 + (2 - kt) * 9 + (2 - kh) * 3 + (2 - kw)];
    else w[kw] = Wt[((int64_t)n * C + c) * 27 + kt * 9 + kh * 3 + kw];
  }
  const float s02 = w[0] + w[2];
  float u[6];
  u[0] = 0.25f * w[0];
  u[1] = (-1.f / 6.f) * (s02 + w[1]);
  u[2] = (-1.f / 6.f) * (s02 - w[1]);
  u[3] = (1.f / 24.f) * w[0] + (1.f / 12.f) * w[1] + (1.f / 6.f) * w[2];
  u[4] = (1.f / 24.f) * w[0] - (1.f / 12.f) * w[1] + (1.f / 6.f) * w[2];
  u[5] = w[2];
  const int64_t blk = ((int64_t)(tap9 * CCH + cc) * NB + nb) * (12 * 64 * 4);
#pragma unroll
  for (int pp = 0; pp < 6; ++pp) U[blk + ((pp * 2 + h) * 64 + nl) * 4 + j] = u[pp];
}

template <int STAGES, bool WPAD, int TG>
static int launch_wino(const SlicConvArgs& a, hipStream_t st, int splits = 1, float* slab = nullptr, int nfull = 0) {
  constexpr size_t ring = (size_t)STAGES * wino_stage_floats(TG) * sizeof(float), epi = (size_t)conv_epi_lds_floats(128, 64, 2 * TG) * sizeof(float);
  constexpr size_t lds = ring > epi ? ring : epi;
  static bool attr_set = false;
  if (!attr_set) {
    SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)conv_wino_kernel<STAGES, WPAD, TG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int64_t tiles = (a.M / a.Ws) * ((a.Ws + 3) / 4);
  const int gx = (int)slic_cdiv(tiles, 32 * TG);               // tile blocks
  const unsigned ny = (unsigned)(a.N / 64);
  if (splits > 1) {
    // the first `nfull` tile blocks run whole (the launch's full dispatch rounds), the rest cut the K loop `splits` ways
    if (nfull > 0) {
      conv_wino_kernel<STAGES, WPAD, TG><<<dim3((unsigned)((nfull + 7) / 8 * 8), ny), dim3(128 * TG), lds, st>>>(a, nullptr, 0, 0, nfull);
      SLIC_LAUNCH_CHECK();
    }
    const int tail = gx - nfull;
    const int ns = 9 * (a.Cs / 8);
    int per = (ns + splits - 1) / splits;
    per = (per + STAGES - 1) / STAGES * STAGES;              // whole ring turns per piece
    const int S = (ns + per - 1) / per;
    conv_wino_kernel<STAGES, WPAD, TG><<<dim3((unsigned)((tail + 7) / 8 * 8), ny, (unsigned)S), dim3(128 * TG), lds, st>>>(a, slab, per, nfull, tail);
    SLIC_LAUNCH_CHECK();
    const int64_t row0 = (int64_t)nfull * 32 * TG * 4, rows = a.M - row0;          // nfull > 0: W % 4 == 0, four GEMM rows per tile
    conv_splitk_finish<128, 64, 2, 2><<<dim3((unsigned)slic_cdiv(rows, 128), ny), dim3(256), 0, st>>>(a, slab, S, (int)(row0 / 128), rows * (int64_t)a.N);
    SLIC_LAUNCH_CHECK();
    return SLIC_OK;
  }
  conv_wino_kernel<STAGES, WPAD, TG><<<dim3((unsigned)((gx + 7) / 8 * 8), ny), dim3(128 * TG), lds, st>>>(a, nullptr, 0, 0, gx);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

// Workgroup shape: 256 threads (TG = 2).  The 512-thread form (TG = 4: one workgroup per CU, the U stage fetched once per 128
// tiles — a quarter less L2 -> LDS traffic) measured no faster at the layer1 shape (218.7 against 220.2 effective TFLOP/s) and slower
// at layer2's (188 against 213: the barrier then spans eight waves); it is not instantiated.
static int launch_wino_pick(const SlicConvArgs& a, hipStream_t st, int splits, float* slab, int nfull = 0) {
  if (a.Ws % 4 != 0) return launch_wino<3, true, 2>(a, st, splits, slab, 0);
  return launch_wino<3, false, 2>(a, st, splits, slab, nfull);
}

// ------------------------------------------------------------------------------------------
// Weight gradient of the same layers by the TRANSPOSED F(4, 3) algorithm: with y = A^T [(G w) . (B^T x)] per W-tile,
//   dL/dw[kw] = sum over tiles of  G^T [ (B^T x) . (A dy) ]
// — the forward's input transform V = B^T x (six points from six pixels), the output transform run backwards Z = A dy (six points
// from the tile's four output gradients), one 64 x 64 product-sum per point, (kt, kh) and tile, and G^T once at the very end: six
// multiplies per (kt, kh, c, n) and tile instead of twelve.
//   Workgroup = one (kt, kh), one 64 c x 64 n block, one slice of the tiles; 2 x 2 waves of 32 c x 32 n, six accumulators each
//   (S_p[c][n], p = 0..5).  The MFMA's k dimension is the TILE: lane (r, h) reads its channel's six pixels of tile 2 ks + h and its
//   column's four gradients with ds_read_b32 (the operands are channel-contiguous in LDS as in HBM; lanes r are consecutive
//   channels: conflict-free, and odd tiles are stored with their channel halves swapped so that the two half-waves hit different
//   banks), transforms them in registers and feeds one MFMA per point.  Stage = 8 tiles: 12 KB of pixels + 8 KB of gradients by
//   LDS-DMA (per-tile records {byte offset of pixel 4 wt, validity bits} from slic_conv_wino_tile_table, loaded a stage ahead),
//   3-stage ring, counted vmcnt, one barrier per stage, the last k-step of a stage multiplied after the next stage's barrier.
//   Slabs [slice][tap9][p][c][n]; conv_wgrad_wino_reduce adds the slices in order, applies G^T and writes the reference layout.
// ------------------------------------------------------------------------------------------
constexpr int WW_TS = 8;                              // tiles per stage
constexpr int WW_X_FLOATS = WW_TS * 6 * 64;
constexpr int WW_Y_FLOATS = WW_TS * 4 * 64;
constexpr int WW_STAGE_FLOATS = WW_X_FLOATS + WW_Y_FLOATS;

// tile_tab[tile] = {pixel index of (b, t, h, 4 wt) (= its GEMM row), bits 0-2: t - 1, t, t + 1 inside; 3-5: h - 1, h, h + 1 inside;
//                   6-11: pixel 4 wt - 1 + a inside the row (a = 0..5); 12-15: output 4 wt + o inside the row (o = 0..3)}
// Tiles per row = ceil(W / 4): on a width that is not a multiple of 4 the last tile of a row is ragged.
__global__ void conv_wino_tile_table_kernel(const SlicConvArgs p, uint2* __restrict__ tab) {
  const int Wq = (p.Ws + 3) >> 2;
  const int64_t tile = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tile >= (p.M / p.Ws) * Wq) return;
  unsigned q = (unsigned)tile;
  const int wt = (int)(q % (unsigned)Wq); q /= (unsigned)Wq;
  const int hh = (int)(q % (unsigned)p.Hs); q /= (unsigned)p.Hs;
  const int tt = (int)(q % (unsigned)p.Ts); q /= (unsigned)p.Ts;
  unsigned mk = 0;
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    mk |= ((unsigned)(tt + o - 1) < (unsigned)p.Ts ? 1u : 0u) << o;
    mk |= ((unsigned)(hh + o - 1) < (unsigned)p.Hs ? 1u : 0u) << (3 + o);
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) mk |= ((unsigned)(4 * wt - 1 + a) < (unsigned)p.Ws ? 1u : 0u) << (6 + a);
#pragma unroll
  for (int o = 0; o < 4; ++o) mk |= (4 * wt + o < p.Ws ? 1u : 0u) << (12 + o);
  tab[tile] = make_uint2((unsigned)((((int64_t)q * p.Ts + tt) * p.Hs + hh) * p.Ws + 4 * wt), mk);
}

template <int STAGES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_wgrad_wino_kernel(const SlicConvArgs p, const float* __restrict__ dy, unsigned dy_bytes, const uint2* __restrict__ tile_tab,
                            float* __restrict__ slab, int tiles_per_split, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the DMAs' LDS bases (M0) then need no v_readfirstlane
  const int wc = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int C = p.Cs, N = p.N, H = p.Hs, W = p.Ws;
  const int CB = C >> 6, NBk = N >> 6;
  const int per_slice = 9 * CB * NBk;
  const int bx = blockIdx.x, gdx = gridDim.x;
  const int v = (bx & 7) * (gdx >> 3) + (bx >> 3);            // XCD-aware: the workgroups of a slice (same tiles) share an L2
  if (v >= per_slice * nsplit) return;
  const int z = v / per_slice;
  int rest = v - z * per_slice;
  const int tap9 = rest / (CB * NBk); rest -= tap9 * (CB * NBk);
  const int cb = rest / NBk, nb = rest - cb * NBk;
  const int kt = tap9 / 3, kh = tap9 - 3 * kt;
  const int Wq = (W + 3) >> 2;
  const int64_t Mt = (p.M / W) * Wq;
  const int64_t tbeg = (int64_t)z * tiles_per_split;
  const int64_t tend = min(tbeg + tiles_per_split, Mt);
  const int nst = tend > tbeg ? (int)((tend - tbeg + WW_TS - 1) / WW_TS) : 0;
  __builtin_amdgcn_s_setprio(SLIC_PRIO_EDGE);
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_tab = __builtin_amdgcn_make_buffer_rsrc((void*)tile_tab, 0, (int)(tend * 8), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFF00u;
  // ---- DMA roles: a thread serves ONE tile of the stage (tl = tid / 32: lanes 0-31 of wave w tile 2 w, lanes 32-63 tile 2 w + 1)
  // with five 16-byte chunks of its 160 (six pixels x 16, then four gradient rows x 16): chunk k = 32 i + (tid & 31).  The DMA
  // writes lane-linear, so the stage image is [wave 4][piece 5][tile parity 2][32 chunks].
  const int tl = tid >> 5, l32 = tid & 31, par = tl & 1;
  unsigned need[5], cst[5], mul[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int k = 32 * i + l32;
    if (i < 3) {
      const int a = k >> 4, j = k & 15;
      const int c4 = j ^ (par << 3);                          // odd tiles: channel halves swapped (bank spread of the two half-waves)
      need[i] = (1u << kt) | (1u << (3 + kh)) | (1u << (6 + a));
      cst[i] = (unsigned)(((((kt - 1) * H + (kh - 1)) * W + (a - 1)) * C + cb * 64 + 4 * c4) * 4);
      mul[i] = (unsigned)(C * 4);
    } else {
      const int kk = k - 96;
      const int o = kk >> 4, j = kk & 15;
      const int n4 = j ^ (par << 3);
      need[i] = 1u << (12 + o);
      cst[i] = (unsigned)((o * N + nb * 64 + 4 * n4) * 4);
      mul[i] = (unsigned)(N * 4);
    }
  }
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  auto load_rec = [&](int s) -> u32x2 {
    const int64_t tile = tbeg + (int64_t)s * WW_TS + tl;
    return __builtin_amdgcn_raw_buffer_load_b64(rs_tab, (int)(tile * 8), 0, 0);          // past the slice: zeros (nothing valid)
  };
  auto issue_piece = [&](int toff, int d, const u32x2 rec) {
#if SLIC_WINO_ABL & 1
    const unsigned off = OOB + 0 * (rec.y + rec.x);
#else
    const unsigned off = ((rec.y & need[d]) == need[d]) ? __umul24(rec.x, mul[d]) + cst[d] : OOB;    // pixel index < 2^24: one v_mad_u32_u24
#endif
    __builtin_amdgcn_raw_ptr_buffer_load_lds(d < 3 ? rs_src : rs_dy,
                                             (__attribute__((address_space(3))) void*)(lds + toff + (wave * 5 + d) * 64 * 4), 16, (int)off, 0, 0, 0);
  };
  f32x16 acc[6];
#pragma unroll
  for (int pp = 0; pp < 6; ++pp)
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[pp][g] = 0.f;
  // prologue: stages 0 .. STAGES - 2 in flight, the record of stage STAGES - 1 loaded
  u32x2 recn;
#pragma unroll
  for (int t = 0; t < STAGES - 1; ++t) {
    const u32x2 rc = load_rec(t);
#pragma unroll
    for (int d = 0; d < 5; ++d) issue_piece(t * WW_STAGE_FLOATS, d, rc);
    asm volatile("" ::: "memory");
  }
  recn = load_rec(STAGES - 1);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_setprio(0);
  // stage image (floats): pixel a of tile 2 ks + h = piece a / 2 of wave ks, parity h, chunk 16 (a % 2) + channel / 4:
  //   h * 128 + 32 (wc ^ h) + r  +  ks * 1280 + (a >> 1) * 256 + (a & 1) * 64;   gradients: the same behind 3 * 256, with wn
  // k-steps are transformed in PAIRS, two floats per VALU instruction (k-step 2 kp in .x, 2 kp + 1 in .y: the two values of a pair
  // come out of LDS 5120 bytes apart, one ds_read2st64_b32): 21 packed instructions per pair instead of ~44 scalar ones — measured on
  // this kernel family, wave time = MFMA cycles + VALU cycles (see the helpers above).
  const f32x2 c2 = {2.f, 2.f}, c4 = {4.f, 4.f}, c5 = {5.f, 5.f}, c8 = {8.f, 8.f};
  const f32x2 zero2 = {0.f, 0.f};
  // k-steps 2 kp (.x) and 2 kp + 1 (.y) of ring slot SL: every offset is an immediate (256-byte units: slot 80, k-step 20, piece 4,
  // pixel parity 1, gradients 12), the two lane addresses are loop constants
  const unsigned lbase = (unsigned)(size_t)((__attribute__((address_space(3))) float*)lds);
  const unsigned xaddr = lbase + (unsigned)(h * 128 + 32 * (wc ^ h) + r) * 4u;
  const unsigned yaddr = lbase + (unsigned)(h * 128 + 32 * (wn ^ h) + r) * 4u;
  const unsigned xaddr2 = xaddr + 2u * WW_STAGE_FLOATS * 4u, yaddr2 = yaddr + 2u * WW_STAGE_FLOATS * 4u;      // slots 2, 3
  auto read_pair = [&](auto sl_, auto kp_, f32x2 (&x)[6], f32x2 (&y)[4]) {
    constexpr int SL = decltype(sl_)::value;
    constexpr int B0 = (SL & 1) * 80 + decltype(kp_)::value * 40;      // the 8-bit offsets reach two slots from a base address
    const unsigned xa = SL < 2 ? xaddr : xaddr2, ya = SL < 2 ? yaddr : yaddr2;
    x[0] = lds_read2st64<B0 + 0, B0 + 20>(xa);
    x[1] = lds_read2st64<B0 + 1, B0 + 21>(xa);
    x[2] = lds_read2st64<B0 + 4, B0 + 24>(xa);
    x[3] = lds_read2st64<B0 + 5, B0 + 25>(xa);
    x[4] = lds_read2st64<B0 + 8, B0 + 28>(xa);
    x[5] = lds_read2st64<B0 + 9, B0 + 29>(xa);
    y[0] = lds_read2st64<B0 + 12, B0 + 32>(ya);
    y[1] = lds_read2st64<B0 + 13, B0 + 33>(ya);
    y[2] = lds_read2st64<B0 + 16, B0 + 36>(ya);
    y[3] = lds_read2st64<B0 + 17, B0 + 37>(ya);
  };
  // V = B^T x (wino_bt6) and the four inner points of Z = A y  (Z = [y0, y0+y1+y2+y3, y0-y1+y2-y3, y0+2y1+4y2+8y3, y0-2y1+4y2-8y3, y3];
  // points 0 and 5 are y0 and y3 themselves); one fenced block closed by the two wait states an MFMA needs behind the
  // (inline-assembly) VALU instruction that wrote its operand
  auto transform_pair = [&](const f32x2 (&x)[6], const f32x2 (&y)[4], f32x2 (&V)[6], f32x2 (&Zi)[4]) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the pair's (inline-assembly) LDS reads
    wino_bt6(x, V, c2, c4, c5);
    const f32x2 e = pk_add(y[0], y[2]), o = pk_add(y[1], y[3]);
    const f32x2 e4 = pk_fma(y[2], c4, y[0]), o4 = pk_fma(y[3], c8, pk_add(y[1], y[1]));
    Zi[0] = pk_add(e, o);
    Zi[1] = pk_sub(e, o);
    Zi[2] = pk_add(e4, o4);
    Zi[3] = pk_sub(e4, o4);
    asm volatile("s_nop 1" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // the six MFMAs of k-step `sel` (0 / 1) of a transformed pair
  auto mfma6 = [&](const f32x2 (&V)[6], const f32x2 (&Zi)[4], const f32x2 (&y)[4], const int sel) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[0][sel], y[0][sel], acc[0], 0, 0, 0);
#pragma unroll
    for (int pp = 1; pp < 5; ++pp) acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[pp][sel], Zi[pp - 1][sel], acc[pp], 0, 0, 0);
    acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[5][sel], y[3][sel], acc[5], 0, 0, 0);
  };
  // the stage's second pair stays in registers across the barrier: its odd k-step is multiplied after the NEXT stage's barrier
  f32x2 Vp[6], Zp[4], yp[4];
#pragma unroll
  for (int pp = 0; pp < 6; ++pp) Vp[pp] = zero2;
#pragma unroll
  for (int q = 0; q < 4; ++q) { Zp[q] = zero2; yp[q] = zero2; }
  constexpr int PER = 6;                                       // VMEM ops per stage: five DMAs + one record load
  auto stage = [&](const int sg, auto sl_) {
    constexpr int sidx = decltype(sl_)::value;
    // stage sg has landed once only the younger ops are outstanding: the record load issued behind its DMAs and the
    // STAGES - 2 stages after it; and this wave's own LDS reads of the previous stage are complete
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(1 + (STAGES - 2) * PER) : "memory");
#if !(SLIC_WINO_ABL & 2)
    __builtin_amdgcn_s_barrier();
#endif
    constexpr int toffn = ((sidx + STAGES - 1) % STAGES) * WW_STAGE_FLOATS;
    f32x2 xa[6], ya[4], xb[6], V[6], Zi[4];
    read_pair(sl_, std::integral_constant<int, 0>{}, xa, ya);
    mfma6(Vp, Zp, yp, 1);                                      // the previous stage's last k-step, under the latency of these reads
    transform_pair(xa, ya, V, Zi);
    read_pair(sl_, std::integral_constant<int, 1>{}, xb, yp);
    mfma6(V, Zi, ya, 0);
    issue_piece(toffn, 0, recn);
    issue_piece(toffn, 1, recn);
    mfma6(V, Zi, ya, 1);
    issue_piece(toffn, 2, recn);
    issue_piece(toffn, 3, recn);
    issue_piece(toffn, 4, recn);
    asm volatile("" ::: "memory");                           // the counted vmcnt relies on this order: five DMAs, then the record
    recn = load_rec(sg + STAGES);
    asm volatile("" ::: "memory");
    transform_pair(xb, yp, Vp, Zp);
    mfma6(Vp, Zp, yp, 0);
  };
  static_assert(STAGES == 3 || STAGES == 4, "the ring is unrolled by hand");
  for (int s0 = 0; s0 < nst; s0 += STAGES) {
    stage(s0, std::integral_constant<int, 0>{});
    stage(s0 + 1, std::integral_constant<int, 1>{});
    stage(s0 + 2, std::integral_constant<int, 2>{});
    if constexpr (STAGES == 4) stage(s0 + 3, std::integral_constant<int, 3>{});
  }
  mfma6(Vp, Zp, yp, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_setprio(SLIC_PRIO_EDGE);
  // slab[z][tap9][p][c][n]
  float* out = slab + ((int64_t)z * 9 + tap9) * 6 * (int64_t)C * N;
  const int n = nb * 64 + 32 * wn + r;
#pragma unroll
  for (int pp = 0; pp < 6; ++pp)
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int c = cb * 64 + 32 * wc + (g & 3) + 8 * (g >> 2) + 4 * h;
      out[((int64_t)pp * C + c) * N + n] = acc[pp][g];
    }
}

// dW[n][c][kt][kh][kw] = sum_p G[p][kw] * (sum over slices z, ascending, of slab[z][tap9][p][c][n])
__global__ __launch_bounds__(256) void conv_wgrad_wino_reduce(const float* __restrict__ slab, int S, int C, int N, float* __restrict__ dW) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t CN = (int64_t)C * N;
  if (e >= 9 * CN) return;
  const int n = (int)(e % N);
  const int c = (int)((e / N) % C);
  const int tap9 = (int)(e / CN);
  const int64_t zs = 9 * 6 * CN;
  float sp[6];
#pragma unroll
  for (int pp = 0; pp < 6; ++pp) {
    const float* q = slab + ((int64_t)tap9 * 6 + pp) * CN + (int64_t)c * N + n;
    float a = 0.f;
    int zi = 0;
    for (; zi + 4 <= S; zi += 4) {
      const float s0 = q[zi * zs], s1 = q[(zi + 1) * zs], s2 = q[(zi + 2) * zs], s3 = q[(zi + 3) * zs];
      a += s0; a += s1; a += s2; a += s3;
    }
    for (; zi < S; ++zi) a += q[zi * zs];
    sp[pp] = a;
  }
  // G^T rows: kw 0: [1/4 -1/6 -1/6 1/24 1/24 0], kw 1: [0 -1/6 1/6 1/12 -1/12 0], kw 2: [0 -1/6 -1/6 1/6 1/6 1]
  const float s12 = sp[1] + sp[2], d12 = sp[2] - sp[1], s34 = sp[3] + sp[4], d34 = sp[3] - sp[4];
  float* o = dW + ((int64_t)n * C + c) * 27 + tap9 * 3;
  o[0] = 0.25f * sp[0] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
  o[1] = (1.f / 6.f) * d12 + (1.f / 12.f) * d34;
  o[2] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + sp[5];
}

static int validate(const SlicConvArgs* a, const char* who) {
  SLIC_REQUIRE(a && a->src && a->tab, "%s: null pointer", who);
  if (a->k_run_len > 0)
    SLIC_REQUIRE(a->Cs > 0 && a->k_run_len % 4 == 0 && a->k_run_px > 0 && a->k_run_px * a->Cs <= a->k_run_len && !a->tap_tab,
                 "%s: W-run operand needs k_run_len %% 4 == 0, k_run_px * Cs <= k_run_len and no tap table", who);
  else
    SLIC_REQUIRE(a->Cs > 0 && a->Cs % 4 == 0, "%s: source channels must be a multiple of 4 (Cs=%d)", who, a->Cs);
  SLIC_REQUIRE(a->nchunks >= 0 && a->nchunks % 8 == 0, "%s: nchunks %% 8 != 0 (%d)", who, a->nchunks);
  SLIC_REQUIRE(a->M > 0 && a->N > 0 && a->Ga > 0 && a->Gb > 0 && a->Gc > 0 && a->Ts > 0 && a->Hs > 0 && a->Ws > 0,
               "%s: bad shape", who);
  SLIC_REQUIRE(a->M % ((int64_t)a->Ga * a->Gb * a->Gc) == 0, "%s: M is not batch * grid", who);
  SLIC_REQUIRE(a->M < (1ll << 31), "%s: M >= 2^31 rows (split the batch)", who);
  if (a->dst) {   // the epilogue addresses dst / addend / mask_src / bwd_z with 32-bit byte offsets
    const int64_t rows = a->dst_strided ? (a->M / ((int64_t)a->Ga * a->Gb * a->Gc)) * a->Da * a->Db * a->Dc : a->M;
    SLIC_REQUIRE(rows > 0 && rows * (int64_t)a->ldo * 4 < (int64_t)0xFFFFFF00u, "%s: dst larger than 4 GiB (split the batch)", who);
  }
  if (a->dst)     // the epilogue moves 16-byte chunks of dst rows
    SLIC_REQUIRE(a->N % 4 == 0 && a->ldo % 4 == 0 && ((uintptr_t)a->dst % 16) == 0 && (!a->addend || ((uintptr_t)a->addend % 16) == 0),
                 "%s: N and ldo must be multiples of 4 and dst / addend 16-byte aligned", who);
  SLIC_REQUIRE(((uintptr_t)a->src % 16) == 0, "%s: src not 16-byte aligned", who);
  SLIC_REQUIRE(a->src_bytes > 0 && a->src_bytes < 0xFFFFFF00u, "%s: src_bytes must be set and < 4 GiB (split the batch)", who);
  return SLIC_OK;
}

template <int BM, int BN, int WM, int WN>
static int launch_gemm(const SlicConvArgs& a, hipStream_t st) {
  constexpr size_t ring = (size_t)2 * (BM + BN) * 32 * sizeof(float), epi = (size_t)conv_epi_lds_floats(BM, BN) * sizeof(float);
  constexpr size_t lds = ring > epi ? ring : epi;
  static bool attr_set = false;
  if (!attr_set) {
    SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)conv_gemm_kernel<BM, BN, WM, WN>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid((unsigned)slic_cdiv(a.M, BM), (unsigned)slic_cdiv(a.N, BN));
  conv_gemm_kernel<BM, BN, WM, WN><<<grid, dim3(256), lds, st>>>(a);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

template <int BM, int BN, int WM, int WN, int STAGES = 2, int KD = 32>
static int launch_gemm_dma(const SlicConvArgs& a, hipStream_t st) {
  constexpr size_t ring = (size_t)STAGES * (BM + BN) * KD * sizeof(float), epi = (size_t)conv_epi_lds_floats(BM, BN) * sizeof(float);
  constexpr size_t lds = ring > epi ? ring : epi;
  static bool attr_set = false;
  if (!attr_set) {
    SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)conv_gemm_dma_kernel<BM, BN, WM, WN, STAGES, KD>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const unsigned gx = (unsigned)slic_cdiv(a.M, BM);
  // XCD-aware order (see the kernel body): the grid is rounded up to a multiple of 8 row blocks
  dim3 grid((gx + 7) / 8 * 8, (unsigned)slic_cdiv(a.N, BN));
  conv_gemm_dma_kernel<BM, BN, WM, WN, STAGES, KD><<<grid, dim3(256), lds, st>>>(a, 1, nullptr, 0);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

template <int BM, int BN, int WM, int WN, int STAGES = 2, int KD = 32>
static int launch_gemm_dma_tail(const SlicConvArgs& a, hipStream_t st, int nfull_rb, int splits, float* slab) {
  constexpr size_t ring = (size_t)STAGES * (BM + BN) * KD * sizeof(float), epi = (size_t)conv_epi_lds_floats(BM, BN) * sizeof(float);
  constexpr size_t lds = ring > epi ? ring : epi;
  static bool attr_set = false;
  if (!attr_set) {
    SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)conv_gemm_dma_tail_kernel<BM, BN, WM, WN, STAGES, KD>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int nrb = (int)slic_cdiv(a.M, BM), ny = (int)slic_cdiv(a.N, BN);
  const int nk = a.nchunks / (KD / 4);
  const int per = (nk + splits - 1) / splits;
  const int S = (nk + per - 1) / per;
  const int gxf = (nfull_rb + 7) / 8 * 8;
  const int tail_rb = nrb - nfull_rb;
  const int64_t zs = (int64_t)tail_rb * BM * a.N;
  const int64_t total = (int64_t)gxf * ny + (int64_t)tail_rb * ny * S;
  conv_gemm_dma_tail_kernel<BM, BN, WM, WN, STAGES, KD><<<dim3((unsigned)total), dim3(256), lds, st>>>(a, slab, per, nfull_rb, gxf, ny,
                                                                                                       S, zs);
  SLIC_LAUNCH_CHECK();
  conv_splitk_finish<BM, BN, WM, WN><<<dim3((unsigned)tail_rb, (unsigned)ny), dim3(256), 0, st>>>(a, slab, S, nfull_rb, zs);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

// Variants of the gather-GEMM (the tile sweeps that chose them are in DESIGN.md; the losers were removed from the library):
//   0   register-staged 64 x 64 tiles — any source channel count (the W-run stem, tiny-channel layers, plain GEMMs)
//   20  LDS-DMA ring, 64 x 64 tiles, 5 workgroups / CU   (source channels % 32 == 0)
//   22  LDS-DMA ring, 128 x 64 tiles, 3 workgroups / CU  (N <= 64 and many rows: layer1)
//   30  Winograd F(4, 3) along W (3 x 3 x 3 / stride 1 / pad 1)
//   31  Winograd F(4, 3) x F(2, 3) over (W, H) (conv_wino2.hip; the same layers where they have many tiles)
extern "C" int slic_conv_tile_m(const SlicConvArgs* a, int variant) {
  // rows per slab row (callers size stat_partial / bwd_partial with it).  Variant 30 on a width that is not a multiple of 4: a
  // block of 128 image rows holds 128 / Wp * W real rows (conv_epilogue_rows, WPAD)
  if (variant == 31) return a ? slic_wino2_full_rows(a) : 512;   // two-dimensional Winograd: the real outputs of 64 tiles (0: not eligible)
  if (variant == 30 && a && a->Ws % 4 != 0) return 128 / ((a->Ws + 3) / 4 * 4) * a->Ws;
  return (variant == 22 || variant == 30) ? 128 : 64;
}

extern "C" int slic_conv_gemm(const SlicConvArgs* a, int variant, void* stream) {
  int rc = validate(a, "slic_conv_gemm");
  if (rc) return rc;
  SLIC_REQUIRE(a->wgt && a->dst && a->ldw % 4 == 0 && a->ldo >= 1, "slic_conv_gemm: bad weight/dst");
  SLIC_REQUIRE(((uintptr_t)a->wgt % 16) == 0, "slic_conv_gemm: wgt not 16-byte aligned");
  SLIC_REQUIRE(a->wgt_bytes > 0 && a->wgt_bytes < 0xFFFFFF00u, "slic_conv_gemm: wgt_bytes must be set and < 4 GiB");
  SLIC_REQUIRE(!a->bwd_partial || (a->bwd_z && a->bwd_mean && a->bwd_invstd && !a->stat_partial),
               "slic_conv_gemm: bwd_partial needs bwd_z, bwd_mean, bwd_invstd (and excludes stat_partial)");
  SLIC_REQUIRE(variant == 0 || variant == 20 || variant == 22 || variant == 30 || variant == 31,
               "slic_conv_gemm: variant must be 0, 20, 22, 30 or 31");
  hipStream_t st = S_(stream);
  if (variant == 31) return slic_conv_wino2_launch(a, st, -1, nullptr, 1);   // Winograd F(4, 3) x F(2, 3) over (W, H): wgt = the operand of slic_pack_weight_wino2
  if (variant == 30) {
    // Winograd F(4, 3) along W: wgt = the operand of slic_pack_weight_wino; 3 x 3 x 3, stride 1, pad 1 geometry only
    SLIC_REQUIRE(a->Cs % 8 == 0 && a->N % 64 == 0 && a->sa == 1 && a->sb == 1 && a->sc == 1 && a->Ga == a->Ts &&
                     a->Gb == a->Hs && a->Gc == a->Ws && !a->dst_strided && !a->bias && !a->k_run_len,
                 "slic_conv_gemm: variant 30 needs a stride-1 same-size geometry, Cs %% 8 == 0, N %% 64 == 0, no bias");
    const int Wp = (a->Ws + 3) / 4 * 4;
    SLIC_REQUIRE(a->Ws % 4 == 0 || 128 % Wp == 0, "slic_conv_gemm: variant 30 needs Ws %% 4 == 0 or a padded width dividing 128 (Ws=%d)", a->Ws);
    SLIC_REQUIRE(((a->Cs / 8) & (a->Cs / 8 - 1)) == 0, "slic_conv_gemm: variant 30 needs Cs / 8 to be a power of two");
    SLIC_REQUIRE((uint64_t)a->wgt_bytes == (uint64_t)9 * a->Cs * a->N * 6 * 4, "slic_conv_gemm: variant 30: wgt_bytes != 9 * Cs * N * 6 floats");
    return launch_wino_pick(*a, st, 1, nullptr);
  }
  if (variant != 0) {
    SLIC_REQUIRE(a->tap_tab && a->Cs % 32 == 0 && a->nchunks * 4 % a->Cs == 0 && a->nchunks * 4 / a->Cs <= 64,
                 "slic_conv_gemm: LDS-DMA variants need tap_tab, source channels %% 32 == 0 and <= 64 taps");
    SlicConvArgs b = *a;
    b.tab = a->tap_tab;
    if (variant == 22) return launch_gemm_dma<128, 64, 2, 2>(b, st);
    return launch_gemm_dma<64, 64, 2, 2>(b, st);
  }
  return launch_gemm<64, 64, 2, 2>(*a, st);
}

template <int BM, int BN, int WM, int WN, int STAGES = 2, int KD = 32>
static int launch_gemm_dma_multi(const SlicConvArgs* a, int n, hipStream_t st) {
  constexpr size_t ring = (size_t)STAGES * (BM + BN) * KD * sizeof(float), epi = (size_t)conv_epi_lds_floats(BM, BN) * sizeof(float);
  constexpr size_t lds = ring > epi ? ring : epi;
  static bool attr_set = false;
  if (!attr_set) {
    SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)conv_gemm_dma_multi_kernel<BM, BN, WM, WN, STAGES, KD>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  constexpr int xcd = 1;
  SlicConvArgsPack pk;
  unsigned gmax = 0;
  for (int i = 0; i < SLIC_CONV_MULTI_MAX; ++i) {
    pk.a[i] = a[i < n ? i : 0];
    pk.a[i].tab = a[i < n ? i : 0].tap_tab;
    unsigned gx = i < n ? (unsigned)slic_cdiv(a[i].M, BM) : 0u;
    if (xcd) gx = (gx + 7) / 8 * 8;
    pk.gx[i] = (int)gx;
    if (gx > gmax) gmax = gx;
  }
  dim3 grid(gmax, (unsigned)slic_cdiv(a[0].N, BN), (unsigned)n);
  conv_gemm_dma_multi_kernel<BM, BN, WM, WN, STAGES, KD><<<grid, dim3(256), lds, st>>>(pk, xcd);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_conv_gemm_multi(const SlicConvArgs* args, int n, int variant, void* stream) {
  SLIC_REQUIRE(args && n >= 1 && n <= SLIC_CONV_MULTI_MAX, "slic_conv_gemm_multi: 1 <= n <= %d", SLIC_CONV_MULTI_MAX);
  SLIC_REQUIRE(variant == 20 || variant == 22, "slic_conv_gemm_multi: variants 20 and 22 only");
  for (int i = 0; i < n; ++i) {
    const SlicConvArgs* a = args + i;
    int rc = validate(a, "slic_conv_gemm_multi");
    if (rc) return rc;
    SLIC_REQUIRE(a->wgt && a->dst && a->ldw % 4 == 0 && a->ldo >= 1 && ((uintptr_t)a->wgt % 16) == 0 && a->wgt_bytes > 0 &&
                 a->wgt_bytes < 0xFFFFFF00u, "slic_conv_gemm_multi: bad weight/dst");
    SLIC_REQUIRE(!a->bwd_partial || (a->bwd_z && a->bwd_mean && a->bwd_invstd && !a->stat_partial),
                 "slic_conv_gemm_multi: bwd_partial needs bwd_z, bwd_mean, bwd_invstd (and excludes stat_partial)");
    SLIC_REQUIRE(a->tap_tab && a->Cs % 32 == 0 && a->nchunks * 4 % a->Cs == 0 && a->nchunks * 4 / a->Cs <= 64,
                 "slic_conv_gemm_multi: LDS-DMA variants need tap_tab, source channels %% 32 == 0 and <= 64 taps");
    SLIC_REQUIRE(a->N == args[0].N, "slic_conv_gemm_multi: every GEMM must have the same N");
  }
  if (variant == 22) return launch_gemm_dma_multi<128, 64, 2, 2>(args, n, S_(stream));
  return launch_gemm_dma_multi<64, 64, 2, 2>(args, n, S_(stream));
}

extern "C" size_t slic_conv_gemm_tailsplit_workspace_bytes(const SlicConvArgs* a, int variant, int nfull_rb, int splits) {
  if (!a || splits < 1 || nfull_rb < 0) return 0;
  if (variant == 31) return slic_conv_wino2_split_workspace_bytes(a, nfull_rb, splits);    // pieces of the tile blocks behind the first nfull_rb
  if (variant == 30) {                                         // pieces hold whole outputs of the rows behind the first nfull_rb 64-tile blocks
    const int64_t row0 = a->Ws % 4 == 0 ? (int64_t)nfull_rb * 256 : 0;
    return row0 < a->M ? slic_align_up((size_t)splits * (a->M - row0) * a->N * sizeof(float), 256) : 0;
  }
  const int BM = variant == 22 ? 128 : 64;
  const int64_t tail_rb = slic_cdiv(a->M, BM) - nfull_rb;
  if (tail_rb <= 0) return 0;
  return slic_align_up((size_t)splits * tail_rb * BM * a->N * sizeof(float), 256);
}

extern "C" int slic_conv_gemm_tailsplit(const SlicConvArgs* a, int variant, int nfull_rb, int splits, void* workspace, void* stream) {
  int rc = validate(a, "slic_conv_gemm_tailsplit");
  if (rc) return rc;
  if (variant == 31) {
    // two-dimensional Winograd: the first nfull_rb 64-tile blocks whole, the blocks behind them with the K loop cut into `splits` even
    // pieces (splits | 3 Cs / 16) + conv_wino2_finish: the partly filled last dispatch round of a launch (layer2 at B = 32: 784
    // workgroups on 256 slots) and launches of few workgroups (layer4: 64 workgroups x 4 pieces, small batches)
    SLIC_REQUIRE(a->wgt && a->dst && nfull_rb >= 0 && (!a->bwd_partial || (a->bwd_z && a->bwd_mean && a->bwd_invstd && !a->stat_partial)),
                 "slic_conv_gemm_tailsplit: variant 31: bad args");
    if (splits <= 1) return slic_conv_wino2_launch(a, S_(stream), -1, nullptr, 1);
    return slic_conv_wino2_launch(a, S_(stream), nfull_rb, (float*)workspace, splits);
  }
  if (variant == 30) {
    // Winograd with the K loop (the 9 x Cs / 8 stages) cut `splits` ways: the few-tile layers (layer4 at B = 32: 112 workgroups of
    // 576 stages).  Pieces write transformed outputs to workspace[piece][M][N]; the finish pass adds them in piece order and runs
    // the epilogue with slab rows of 128 GEMM rows (also on a ragged width).
    // nfull_rb > 0 (widths that are multiples of 4): the first nfull_rb 64-tile blocks — the launch's full dispatch rounds — run whole,
    // only the blocks behind them cut their K loop (layer2 at B = 32: 1568 workgroups on 512 slots, 218 -> 236 TFLOP/s without the tail)
    const int64_t gx = slic_cdiv((a->M / a->Ws) * ((a->Ws + 3) / 4), 64);
    if (splits <= 1 || nfull_rb >= gx) return slic_conv_gemm(a, variant, stream);
    SLIC_REQUIRE(nfull_rb >= 0 && workspace && (nfull_rb == 0 || a->Ws % 4 == 0),
                 "slic_conv_gemm_tailsplit: variant 30: whole blocks in front of the split need a width that is a multiple of 4");
    SLIC_REQUIRE(a->wgt && a->dst && a->Cs % 8 == 0 && a->N % 64 == 0 && a->sa == 1 && a->sb == 1 && a->sc == 1 && a->Ga == a->Ts &&
                     a->Gb == a->Hs && a->Gc == a->Ws && !a->dst_strided && !a->bias && !a->k_run_len && ((a->Cs / 8) & (a->Cs / 8 - 1)) == 0,
                 "slic_conv_gemm_tailsplit: variant 30 geometry (see slic_conv_gemm)");
    SLIC_REQUIRE((uint64_t)a->wgt_bytes == (uint64_t)9 * a->Cs * a->N * 6 * 4, "slic_conv_gemm_tailsplit: variant 30: wgt_bytes != 9 * Cs * N * 6 floats");
    SLIC_REQUIRE(!a->bwd_partial || (a->bwd_z && a->bwd_mean && a->bwd_invstd && !a->stat_partial),
                 "slic_conv_gemm_tailsplit: bwd_partial needs bwd_z, bwd_mean, bwd_invstd (and excludes stat_partial)");
    return launch_wino_pick(*a, S_(stream), splits, (float*)workspace, nfull_rb);
  }
  SLIC_REQUIRE(variant == 20 || variant == 22, "slic_conv_gemm_tailsplit: variants 20, 22 (tail split) and 30 (split-K) only");
  const int BM = variant == 22 ? 128 : 64;
  const int64_t nrb = slic_cdiv(a->M, BM);
  if (splits <= 1 || nfull_rb >= nrb) return slic_conv_gemm(a, variant, stream);
  SLIC_REQUIRE(nfull_rb >= 0, "slic_conv_gemm_tailsplit: nfull_rb < 0");
  SLIC_REQUIRE(a->wgt && a->dst && workspace && a->ldw % 4 == 0 && a->ldo >= 1, "slic_conv_gemm_tailsplit: bad weight/dst/workspace");
  SLIC_REQUIRE(((uintptr_t)a->wgt % 16) == 0 && a->wgt_bytes > 0 && a->wgt_bytes < 0xFFFFFF00u, "slic_conv_gemm_tailsplit: bad wgt");
  SLIC_REQUIRE(!a->bwd_partial || (a->bwd_z && a->bwd_mean && a->bwd_invstd && !a->stat_partial),
               "slic_conv_gemm_tailsplit: bwd_partial needs bwd_z, bwd_mean, bwd_invstd (and excludes stat_partial)");
  SLIC_REQUIRE(a->tap_tab && a->Cs % 32 == 0 && a->nchunks * 4 % a->Cs == 0 && a->nchunks * 4 / a->Cs <= 64,
               "slic_conv_gemm_tailsplit: LDS-DMA variants need tap_tab, source channels %% 32 == 0 and <= 64 taps");
  const int64_t ny = slic_cdiv(a->N, 64);
  SLIC_REQUIRE(((nfull_rb + 7) / 8 * 8 + (nrb - nfull_rb) * (int64_t)splits) * ny < (1ll << 31), "slic_conv_gemm_tailsplit: grid too large");
  SlicConvArgs b = *a;
  b.tab = a->tap_tab;
  if (variant == 22) return launch_gemm_dma_tail<128, 64, 2, 2>(b, S_(stream), nfull_rb, splits, (float*)workspace);
  return launch_gemm_dma_tail<64, 64, 2, 2>(b, S_(stream), nfull_rb, splits, (float*)workspace);
}

extern "C" size_t slic_conv_wgrad_workspace_bytes(const SlicConvArgs* a, int splits) {
  if (!a || splits < 1) return 0;
  return slic_align_up((size_t)splits * a->N * a->nchunks * 4 * sizeof(float), 256);
}

extern "C" int slic_conv_wgrad(const SlicConvArgs* a, const float* dy, int ldy, int splits, int C,
                               int ntaps, float* dW, void* workspace, void* stream) {
  int rc = validate(a, "slic_conv_wgrad");
  if (rc) return rc;
  SLIC_REQUIRE(dy && dW && workspace && splits >= 1 && ldy % 4 == 0 && C > 0 && ntaps > 0 && a->N % 4 == 0,
               "slic_conv_wgrad: bad args (N %% 4 == 0 required)");
  const int RL = a->k_run_len > 0 ? a->k_run_len : a->Cs, PPR = a->k_run_len > 0 ? a->k_run_px : 1;
  SLIC_REQUIRE((int64_t)slic_cdiv(ntaps, PPR) * RL <= (int64_t)a->nchunks * 4, "slic_conv_wgrad: table shorter than the taps");
  hipStream_t st = S_(stream);
  float* slab = (float*)workspace;
  int64_t per = slic_cdiv(a->M, splits);
  per = slic_cdiv(per, 32) * 32;
  const int S = (int)slic_cdiv(a->M, per);
  const int Kp = a->nchunks * 4;
  const int64_t dyb = a->M * (int64_t)ldy * 4;
  SLIC_REQUIRE(dyb < (int64_t)0xFFFFFF00u, "slic_conv_wgrad: dy larger than 4 GiB (split the batch)");
  SLIC_REQUIRE(a->M < (int64_t)0x7FFFFFFF, "slic_conv_wgrad: M must fit 31 bits");
  // LDS-DMA kernel, 128 x 64 output tile, 2-stage ring, interleaved issue; with args->row_tab the rows come from the table
  int rc2;
  if (a->row_tab) rc2 = launch_wgrad_dma<2, 2, true, true>(*a, dy, ldy, (unsigned)dyb, slab, (int)per, S, st);
  else rc2 = launch_wgrad_dma<2, 2, true, false>(*a, dy, ldy, (unsigned)dyb, slab, (int)per, S, st);
  if (rc2) return rc2;
  SLIC_LAUNCH_CHECK();
  const int64_t tot = (int64_t)a->N * Kp;
  if (a->k_run_len > 0)
    conv_wgrad_reduce<true><<<dim3((unsigned)slic_cdiv(tot / 4, 256)), dim3(256), 0, st>>>(slab, S, a->N, Kp, a->Cs, C, ntaps, RL, PPR, dW);
  else
    conv_wgrad_reduce<false><<<dim3((unsigned)slic_cdiv(tot / 4, 256)), dim3(256), 0, st>>>(slab, S, a->N, Kp, a->Cs, C, ntaps, RL, PPR, dW);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

static int64_t wino_tiles(const SlicConvArgs* a) { return (a->M / a->Ws) * ((a->Ws + 3) / 4); }

static int wino_wgrad_plan(const SlicConvArgs* a, int splits, int* tps, int* S) {
  const int64_t Mt = wino_tiles(a);
  int64_t per = slic_cdiv(Mt, splits < 1 ? 1 : splits);
  per = slic_cdiv(per, WW_TS) * WW_TS;
  *tps = (int)per;
  *S = (int)slic_cdiv(Mt, per);
  return 0;
}

extern "C" size_t slic_conv_wgrad_wino_workspace_bytes(const SlicConvArgs* a, int splits) {
  if (!a || a->M <= 0) return 0;
  int tps, S;
  wino_wgrad_plan(a, splits, &tps, &S);
  return slic_align_up((size_t)S * 9 * 6 * a->Cs * a->N * sizeof(float), 256);
}

extern "C" int slic_conv_wino_tile_table(const SlicConvArgs* a, uint32_t* tile_tab, void* stream) {
  SLIC_REQUIRE(a && tile_tab && a->M > 0 && a->Ws > 0 && a->Ga == a->Ts && a->Gb == a->Hs && a->Gc == a->Ws && a->Cs > 0,
               "slic_conv_wino_tile_table: needs a stride-1 same-size geometry");
  SLIC_REQUIRE(a->M * (int64_t)a->Cs * 4 < (int64_t)0xFFFFFF00u, "slic_conv_wino_tile_table: source larger than 4 GiB");
  conv_wino_tile_table_kernel<<<dim3((unsigned)slic_cdiv(wino_tiles(a), 256)), dim3(256), 0, S_(stream)>>>(*a, (uint2*)tile_tab);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_conv_wgrad_wino(const SlicConvArgs* a, const float* dy, int splits, const uint32_t* tile_tab, float* dW,
                                    void* workspace, void* stream) {
  int rc = validate(a, "slic_conv_wgrad_wino");
  if (rc) return rc;
  SLIC_REQUIRE(dy && dW && workspace && tile_tab && splits >= 1, "slic_conv_wgrad_wino: bad args");
  SLIC_REQUIRE(a->Cs % 64 == 0 && a->N % 64 == 0 && a->sa == 1 && a->sb == 1 && a->sc == 1 && a->Ga == a->Ts &&
                   a->Gb == a->Hs && a->Gc == a->Ws && !a->k_run_len,
               "slic_conv_wgrad_wino: needs a 3x3x3 stride-1 same-size geometry, Cs %% 64 == 0, N %% 64 == 0");
  const int64_t dyb = a->M * (int64_t)a->N * 4;
  SLIC_REQUIRE(dyb < (int64_t)0xFFFFFF00u, "slic_conv_wgrad_wino: dy larger than 4 GiB (split the batch)");
  int tps, S;
  wino_wgrad_plan(a, splits, &tps, &S);
  hipStream_t st = S_(stream);
  constexpr int STAGES = 3;          // a 4-stage ring (80 KB, still two workgroups per CU) measured equal: the loss is not prefetch depth
  constexpr size_t lds = (size_t)STAGES * WW_STAGE_FLOATS * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)conv_wgrad_wino_kernel<STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int64_t total = (int64_t)9 * (a->Cs / 64) * (a->N / 64) * S;
  SLIC_REQUIRE(total < (1ll << 30), "slic_conv_wgrad_wino: grid too large");
  const unsigned gx = (unsigned)((total + 7) / 8 * 8);
  conv_wgrad_wino_kernel<STAGES><<<dim3(gx), dim3(256), lds, st>>>(*a, dy, (unsigned)dyb, (const uint2*)tile_tab, (float*)workspace, tps, S);
  SLIC_LAUNCH_CHECK();
  const int64_t tot = (int64_t)9 * a->Cs * a->N;
  conv_wgrad_wino_reduce<<<dim3((unsigned)slic_cdiv(tot, 256)), dim3(256), 0, st>>>((const float*)workspace, S, a->Cs, a->N, dW);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_conv_row_table(const SlicConvArgs* a, uint32_t* row_tab, void* stream) {
  SLIC_REQUIRE(a && row_tab && a->M > 0 && a->M < (int64_t)0x7FFFFFFF && a->Ga > 0 && a->Gb > 0 && a->Gc > 0 && a->Cs > 0,
               "slic_conv_row_table: bad args");
  SLIC_REQUIRE(a->M % ((int64_t)a->Ga * a->Gb * a->Gc) == 0, "slic_conv_row_table: M is not a multiple of Ga*Gb*Gc");
  conv_row_table_kernel<<<dim3((unsigned)slic_cdiv(a->M, 256)), dim3(256), 0, S_(stream)>>>(*a, (uint2*)row_tab);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_pack_weight_fwd(const float* W, int N, int C, int ntaps, int Cs, int Kp, float* Wp,
                                    void* stream) {
  SLIC_REQUIRE(W && Wp && N > 0 && C > 0 && ntaps > 0 && Cs >= C && Kp >= ntaps * Cs, "slic_pack_weight_fwd: bad args");
  int CH = 64;                                             // channels per tile: 64, fewer when the taps are many (48 KiB of LDS)
  if ((size_t)CH * (ntaps | 1) * sizeof(float) > 48 * 1024) CH = (int)(12288 / (ntaps | 1));
  SLIC_REQUIRE(CH >= 1, "slic_pack_weight_fwd: %d taps exceed the LDS tile", ntaps);
  const size_t lds = (size_t)CH * (ntaps | 1) * sizeof(float);
  const dim3 grid((unsigned)N, (unsigned)slic_cdiv(C, CH));
  if (ntaps == 27) pack_w_fwd<27><<<grid, dim3(256), lds, S_(stream)>>>(W, N, C, ntaps, Cs, Kp, CH, Wp);
  else if (ntaps == 1) pack_w_fwd<1><<<grid, dim3(256), lds, S_(stream)>>>(W, N, C, ntaps, Cs, Kp, CH, Wp);
  else pack_w_fwd<0><<<grid, dim3(256), lds, S_(stream)>>>(W, N, C, ntaps, Cs, Kp, CH, Wp);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_pack_weight_fwd_runs(const float* W, int N, int C, int ntaps, int run_len, int run_px, int Kp, float* Wp,
                                         void* stream) {
  SLIC_REQUIRE(W && Wp && N > 0 && C > 0 && ntaps > 0 && run_len > 0 && run_px > 0 && run_px * C <= run_len &&
               (int64_t)slic_cdiv(ntaps, run_px) * run_len <= Kp, "slic_pack_weight_fwd_runs: bad args");
  const int64_t tot = (int64_t)N * Kp;
  pack_w_fwd_runs<<<dim3((unsigned)slic_cdiv(tot, 256)), dim3(256), 0, S_(stream)>>>(W, N, C, ntaps, run_len, run_px, Kp, Wp);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_pack_weight_wino(const float* W, int N, int C, int dgrad, float* U, void* stream) {
  SLIC_REQUIRE(W && U && N > 0 && C > 0, "slic_pack_weight_wino: bad args");
  const int N_ = dgrad ? C : N, C_ = dgrad ? N : C;
  SLIC_REQUIRE(N_ % 64 == 0 && C_ % 8 == 0, "slic_pack_weight_wino: needs output channels %% 64 == 0 and reduction channels %% 8 == 0");
  const int64_t tot = (int64_t)9 * C_ * N_;
  pack_w_wino<<<dim3((unsigned)slic_cdiv(tot, 256)), dim3(256), 0, S_(stream)>>>(W, N, C, dgrad, U);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_pack_weight_dgrad(const float* W, int N, int C, int ntaps, int Cs, int Kd, float* Wd,
                                      void* stream) {
  SLIC_REQUIRE(W && Wd && N > 0 && C > 0 && ntaps > 0 && Cs >= C && Kd >= ntaps * N, "slic_pack_weight_dgrad: bad args");
  const int R = C * ntaps;
  if (N % 4 != 0 || R % 4 != 0 || ((uintptr_t)W % 16) != 0 || ((uintptr_t)Wd % 16) != 0 || Kd % 4 != 0) {
    const int64_t tot = (int64_t)R * N;
    pack_w_dgrad_any<<<dim3((unsigned)slic_cdiv(tot, 256)), dim3(256), 0, S_(stream)>>>(W, N, C, ntaps, Kd, Wd);
    SLIC_LAUNCH_CHECK();
    return SLIC_OK;
  }
  pack_w_dgrad<<<dim3((unsigned)slic_cdiv(R, 64), (unsigned)slic_cdiv(N, 64)), dim3(256), 0, S_(stream)>>>(W, N, R, ntaps, Kd, Wd);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_ncdhw_to_ndhwc(const float* x, int B, int C, int64_t S, int Cp, float* y, void* stream) {
  SLIC_REQUIRE(x && y && B > 0 && C > 0 && S > 0 && Cp >= C, "slic_ncdhw_to_ndhwc: bad args");
  const int64_t tot = (int64_t)B * S;
  ncdhw_to_ndhwc<<<dim3((unsigned)slic_cdiv(tot, 256)), dim3(256), 0, S_(stream)>>>(x, B, C, S, Cp, y);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_ncdhw_to_ndhwc_wpad(const float* x, int B, int C, int64_t R, int W, int pad_left, int Wp, float* y, void* stream) {
  SLIC_REQUIRE(x && y && B > 0 && C > 0 && R > 0 && W > 0 && pad_left >= 0 && Wp >= W + pad_left, "slic_ncdhw_to_ndhwc_wpad: bad args");
  const int64_t tot = (int64_t)B * R * Wp;
  ncdhw_to_ndhwc_wpad<<<dim3((unsigned)slic_cdiv(tot, 256)), dim3(256), 0, S_(stream)>>>(x, B, C, R, W, pad_left, Wp, y);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
