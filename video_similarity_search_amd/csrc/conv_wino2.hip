// Two-dimensional Winograd for the 3 x 3 x 3, stride-1, pad-1 convolutions: F(4, 3) along W x F(2, 3) along H.
//
// Replaces the cuDNN calls behind nn.Conv3d of the BasicBlock convolutions (/root/reference/models/resnet.py:11-17, 41-57) —
// forward and, with the flipped / transposed operand, data gradient — on the layers with many tiles (layer1 / layer2 / layer3).
// A tile is 2 output rows x 4 output columns; per kt and channel its 4 x 6 input patch d becomes 24 points V = Bh^T d Bw, each
// point is its own fp32-MFMA GEMM against U = Gh w Gw^T (packed once per step by pack_w_wino2), and the 8 outputs are
// Y = Ah^T M Aw.  24 multiplies per (kt, c, n) and tile where the direct form has 72 and the one-dimensional F(4, 3) kernel 36:
// exact fp32 arithmetic (v_mfma_f32_32x32x2_f32), a third of the direct form's matrix work.
//
// Why two dimensions and not a leaner one-dimensional kernel: on gfx950 a wave's vector and vector-memory instructions do not run
// beside fp32 MFMAs of the same SIMD — the fp32 MFMA holds the vector issue for its 64 cycles (scripts/micro/mfma_valu_coexec.hip:
// SQ_VALU_MFMA_COEXEC_CYCLES = 0 in every build, +3-6 cycles per vector instruction, ~25 per LDS-DMA piece) — so the one-dimensional
// kernel's 0.73-0.75 of the pipe is within a few points of what its instruction mix allows, and the remaining lever is fewer MFMAs.
//
// Workgroup = 512 threads = 8 waves, 64 tiles x 64 n.  The 24 accumulator sets do not fit one wave, so the POINTS are split:
// wave (j, th) owns H-point j (0..3) of tiles 32 th .. 32 th + 31 for all 64 columns: 6 W-points x 2 column halves = 12
// accumulators of 16 registers.  Its H-transform needs TWO of the patch's four rows (Bh^T rows: d0 - d2, d1 + d2, d2 - d1,
// d1 - d3; row 2 is taken as d1 - d2 with U negated), one packed op per pixel, then the F(4, 3) input transform along W:
// 18 packed instructions per stage and wave for 24 MFMAs.
// K loop = 3 C / 4 stages of 4 channels, kt inner (double stage d of 8 channels = (channel group d / 3, kt = d % 3)).  Raw pixels come
// a DOUBLE stage at a time — 48 KB [pixel 24][slot 2][tile half 2][channel half 2][tile 32][4 ch], two slots interleaved per patch
// pixel, 32-byte pieces of a cache line per lane pair — and U a stage at a time — 24 KB [j 4][p 6][column half 2][channel pair 2][n 32][2 ch], two slots — both by
// LDS-DMA in the order the lanes read them (lane (r, hh) of the MFMA reads channels 2 hh, 2 hh + 1 of tile / column r with ONE
// ds_read_b64 per pixel / point: element e goes to MFMA e); 144 KB of rings + 16 KB of per-thread piece offsets: one workgroup per CU,
// two waves per SIMD, counted vmcnt, one barrier per stage, kt and the channel group in the DMA's scalar offset.
// Epilogue: every wave applies Aw^T to its own accumulators (4 W-outputs), the four H-points of a tile are written side by side to
// LDS buf[j 4][tile 64][col 4][n 32] one column half at a time, and w2_epilogue combines them in a fixed order (row 0 = (Y0 + Y1) + Y2,
// row 1 = (Y1 - Y2) - Y3) and finishes the rows (store / affine / addend / ReLU mask / BatchNorm partials as conv_epilogue_rows
// computes them, one slab row per block).  Few-workgroup launches and partly filled last rounds cut the K loop into even pieces
// (slab + conv_wino2_finish).
#include "common.h"
#include "conv_epilogue.h"
#include "conv_internal.h"
#include "wino_common.h"
#include <type_traits>
#include <stdlib.h>

#ifndef SLIC_PRIO_EDGE
#define SLIC_PRIO_EDGE 3
#endif
#ifndef SLIC_W2_ABL
#define SLIC_W2_ABL 0     // diagnostic builds only (scripts/r4/ab_wino2.sh; wrong results, right timing): 1 = DMAs out of range (both kernels), 2 = no stage
                          // barrier (both), 32 = forward without its epilogue, 256 = forward without the row-major half of its epilogue,
                          // 512 = the pixel DMAs of a ROW-IMAGE source ([row][C/8][W][8] planes: 16 image rows x 2 KB per double stage as 32 whole
                          // 1 KB runs instead of 48 gathers of 32-byte pieces — what a planar activation layout would fetch; scripts/r5/ab_planar.sh),
                          // 1024 / 2048 = the forward's BatchNorm statistics without their second moment / skipped (scripts/r5/ab_epilogue.sh: ~2 % of a launch)
                          // 4096 = the weight gradient without its slice-sum and Gh^T .. Gw passes (scripts/r5/ab_wgrad_passes.sh: what they cost a STEP)
                          // 16384 / 32768 (round 6, scripts/r6/ab_taxis.sh: the ceiling of a THIRD Winograd axis, F(2,3) along T — 2/9 of the direct
                          // form's multiplies instead of 1/3): the K loop of two of the three kt (the MFMA count of a three-dimensional kernel at today's
                          // L2 -> LDS bytes per MFMA) / every DMA piece fetched twice (at the 2 x bytes per MFMA its 96-point tiles would need)
#endif
// In-kernel stamps of the workgroup's phases (s_memrealtime, 100 MHz) + the CU it ran on: ONLY in the diagnostic build of
// scripts/r6/stamps_wino2.py (-DSLIC_W2_STAMPS, csrc/_exp/); the shipped library has none.
#ifdef SLIC_W2_STAMPS
__device__ unsigned long long* slic_w2_stamps_buf = nullptr;     // [workgroups][16]
extern "C" int slic_debug_set_w2_stamps(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(slic_w2_stamps_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#define W2_STAMP(slot)                                                                                                         \
  do {                                                                                                                         \
    if (threadIdx.x == 0 && slic_w2_stamps_buf)                                                                                \
      slic_w2_stamps_buf[(size_t)(blockIdx.x + gridDim.x * blockIdx.y) * 16 + (slot)] = __builtin_amdgcn_s_memrealtime();      \
  } while (0)
#define W2_STAMP_ID()                                                                                                          \
  do {                                                                                                                         \
    if (threadIdx.x == 0 && slic_w2_stamps_buf)                                                                                \
      slic_w2_stamps_buf[(size_t)(blockIdx.x + gridDim.x * blockIdx.y) * 16] =                                                 \
          ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);                      \
  } while (0)
// persistent kernel: [workgroup][item 16][slot 8]; slot 0 of item 0 carries the CU id
#define W2P_STAMP(item, slot)                                                                                                  \
  do {                                                                                                                         \
    if (threadIdx.x == 0 && slic_w2_stamps_buf && (item) < 16)                                                                 \
      slic_w2_stamps_buf[((size_t)blockIdx.x * 16 + (item)) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();                  \
  } while (0)
#define W2P_STAMP_ID()                                                                                                         \
  do {                                                                                                                         \
    if (threadIdx.x == 0 && slic_w2_stamps_buf)                                                                                \
      slic_w2_stamps_buf[(size_t)blockIdx.x * 16 * 8 + 7] =                                                                    \
          ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);                      \
  } while (0)
#else
#define W2_STAMP(slot)
#define W2_STAMP_ID()
#define W2P_STAMP(item, slot)
#define W2P_STAMP_ID()
#endif
#ifndef SLIC_W2_UAUX
#define SLIC_W2_UAUX 0    // cache-policy bits of the U / pixel DMAs (experiments: 1 = sc0, 2 = nt, 16 = sc1)
#endif
#ifndef SLIC_W2_PXORD
#define SLIC_W2_PXORD 2    // issue order of a wave's six pixel pieces of a double stage (piece i: patch pixels 4 i + wave / 2; i and i + 3 are patch rows a, a + 2,
                           // the rows a tile shares with the tile below / above it).  2 = {0, 3, 2, 5, 1, 4}: the two rows back to back, so that the second
                           // finds the lines in L1 — +1.3-1.8 % on the layer1 / layer2 launches, 861.9 -> 866.1 clips/s (interleaved); reversed order -8 %;
                           // a piece-to-wave mapping that also keeps the columns a tile shares with its neighbour in one wave measured equal
#endif
#if SLIC_W2_PXORD == 1
#define SLIC_W2_PXORDER {0, 3, 1, 4, 2, 5}
#elif SLIC_W2_PXORD == 2
#define SLIC_W2_PXORDER {0, 3, 2, 5, 1, 4}
#elif SLIC_W2_PXORD == 3
#define SLIC_W2_PXORDER {5, 4, 3, 2, 1, 0}
#else
#define SLIC_W2_PXORDER {0, 1, 2, 3, 4, 5}
#endif
#ifndef SLIC_W2_PAUX
#define SLIC_W2_PAUX 0
#endif
constexpr int W2_PX_FLOATS = 24 * 64 * 8;                      // pixel image of a DOUBLE stage (8 channels): 48 KB
constexpr int W2_U_FLOATS = 24 * 64 * 4;                       // U block of a stage (4 channels): 24 KB
constexpr int W2_RING_FLOATS = 2 * W2_PX_FLOATS + 2 * W2_U_FLOATS;   // two pixel slots + two U slots = 144 KB


constexpr int W2_TREC_FLOATS = 4 * 64 * 4 * 32 + 1024;         // LDS float offset of the 64 tile records, behind buf and the reduction rows

// records of the workgroup's 64 tiles for the epilogue: {GEMM row of the tile's output (0, 0); bit 0: the tile exists, bit 1: its second
// row is inside the frame, bits 2-5: column o is inside}.  Called by threads 0-63 once the K loop's ring is dead; a barrier follows.
__device__ __forceinline__ void w2_tile_records(const SlicConvArgs& p, float* lds, int64_t tile0, int tid) {
  const int H = p.Hs, W = p.Ws;
  const int Wq = (W + 3) >> 2, Hq = (H + 1) >> 1;
  const int64_t Mt = (p.M / ((int64_t)H * W)) * Hq * Wq;
  const int64_t tl = tile0 + tid;
  unsigned q = (unsigned)(tl < Mt ? tl : 0);
  const int wt = (int)(q % (unsigned)Wq); q /= (unsigned)Wq;
  const int h2 = (int)(q % (unsigned)Hq); q /= (unsigned)Hq;       // q = frame (b, t)
  unsigned bits = tl < Mt ? 1u : 0u;
  bits |= (2 * h2 + 1 < H) ? 2u : 0u;
#pragma unroll
  for (int o = 0; o < 4; ++o) bits |= (4 * wt + o < W) ? (4u << o) : 0u;
  ((uint2*)(lds + W2_TREC_FLOATS))[tid] = make_uint2((q * (unsigned)H + 2u * (unsigned)h2) * (unsigned)W + 4u * (unsigned)wt, bits);
}

// Epilogue of the two-dimensional kernel, one call per column half nh.  The four H-point waves of a tile have written their W-outputs
// Yw_j to LDS side by side, buf[j 4][tile 64][col o 4][n 32] (no read-modify-write: 128 accumulate steps per wave through LDS ran at one
// LDS round trip each and cost 13 % of the kernel); this pass reads, for output row (tile, hp, o), the three contributions it is made
// of — Ah^T = [1 1 1 0; 0 1 -1 -1]: row 0 = (Y0 + Y1) + Y2, row 1 = (Y1 - Y2) - Y3 — and does what conv_epilogue_rows does with an
// image row: store / affine / addend / ReLU-backward mask / ReLU / BatchNorm partials (same formulas, fixed reduction order: rows
// ascending in a thread, xor butterfly over a wave's row groups, the eight waves ascending; one slab row per block).  Thread = (row
// group rr = tid / 8, 16-byte chunk cq = tid % 8), 64 rows = 8 tiles per pass, 8 passes.  The vector work is kept small — every
// vector instruction stops the matrix pipe of its SIMD, and with ONE workgroup per CU nothing else runs meanwhile: the tile's
// coordinates are decoded once and stepped (no division per row), absent operands are not loaded (workgroup-uniform branches), the
// combined values stay in registers for the second statistics pass.
// LOADS = the pass reads optional operands from memory (addend, ReLU-backward mask, the BatchNorm z): those loads of ALL eight passes are
// issued first, in one batch, and the passes then consume them — one memory round trip per call instead of one per pass (with one
// workgroup per CU nothing else covers that latency).  The forward of a training step (statistics only) takes the instantiation without.
template <bool LOADS>
__device__ __forceinline__ void w2_epilogue_impl(const SlicConvArgs& p, float* lds, int64_t tile0, int n0h, int tid, int full_rows) {
  constexpr int BNH = 32, CPR = 8, NW = 8, NPASS = 8;          // 64 rows per pass
  constexpr int JSTRIDE = 64 * 4 * BNH;                        // floats between buf[j] and buf[j + 1]
  const int64_t mblk = tile0 >> 6;
  const int W = p.Ws;
  float* red1 = lds + 4 * JSTRIDE;
  float* red2 = red1 + NW * BNH;
  float* bmean = red2 + NW * BNH;
  const int ewave = tid >> 6, elane = tid & 63;
  auto wave_rows_sum = [&](f32x4 v) {
#pragma unroll
    for (int off = CPR; off < 64; off <<= 1) {
      f32x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = __shfl_xor(v[c], off);
      v += o;
    }
    return v;
  };
  const bool want_stats = p.stat_partial != nullptr, want_bwd = p.bwd_partial != nullptr;
  const bool has_add = p.addend != nullptr, has_mask = p.mask_src != nullptr, has_bz = p.bwd_z != nullptr, do_relu = p.relu != 0;
  const bool has_affine = p.scale != nullptr || p.shift != nullptr;
  constexpr unsigned OOBE = 0xFFFFFF00u;
  const unsigned dst_bytes = (unsigned)(((p.M - 1) * (int64_t)p.ldo + p.N) * 4);
  const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc((void*)p.dst, 0, dst_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc((void*)p.addend, 0, has_add ? dst_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_msk = __builtin_amdgcn_make_buffer_rsrc((void*)p.mask_src, 0, has_mask ? dst_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bz = __builtin_amdgcn_make_buffer_rsrc((void*)p.bwd_z, 0, has_bz ? dst_bytes : 0, 0x00020000);
  const int cq = tid & 7, rr = tid >> 3;
  const int n = n0h + cq * 4;
  const bool nv = n < p.N;
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f}, bmu = sh, bis = sh;
  if (nv) {
    if (p.scale) sc = *(const f32x4*)(p.scale + n);
    if (p.shift) sh = *(const f32x4*)(p.shift + n);
    if (want_bwd) { bmu = *(const f32x4*)(p.bwd_mean + n); bis = *(const f32x4*)(p.bwd_invstd + n); }
  }
  // this thread's rows: tile tile0 + 8 ps + rr / 8, row hp, column o.  The tiles' records {GEMM row of output (0, 0), validity bits}
  // wait in LDS (w2_tile_records: 64 threads decoded them once): a pass reads one — no division, no loop, so the compiler is free to
  // hoist the passes' global loads (mask, z, addend) above one another instead of paying one memory round trip per pass
  const int hp = (rr >> 2) & 1, o = rr & 3;
  const float sgn = hp ? -1.f : 1.f;
  const float* src = lds + hp * JSTRIDE + (((rr >> 3) * 4 + o) * BNH + cq * 4);      // + ps * 8 tiles; contributions j = hp, hp + 1, hp + 2
  const uint2* trec = (const uint2*)(lds + W2_TREC_FLOATS) + (rr >> 3);                // + ps * 8
  const unsigned need = 1u | (hp ? 2u : 0u) | (4u << o);
  unsigned okmask = 0;
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1, fs = s1;
  f32x4 keep[NPASS];
  unsigned offs[NPASS];
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const uint2 tr = trec[ps * 8];
    const bool ok = (tr.y & need) == need && nv;
    okmask |= (ok ? 1u : 0u) << ps;
    const unsigned m = tr.x + (unsigned)(hp * W + o);
    offs[ps] = ok ? (m * (unsigned)p.ldo + (unsigned)n) * 4u : OOBE;
  }
  // absent operands have resources of size zero: their loads return zeros at once (and are not issued at all without LOADS)
  // (in two batches of four passes: 48 registers of loads in flight — all eight at once spilled)
  constexpr int LB = 4;
  f32x4 ldadd[LOADS ? LB : 1], ldmsk[LOADS ? LB : 1], ldz[LOADS ? LB : 1];
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    if constexpr (LOADS) {
      if (ps % LB == 0) {
#pragma unroll
        for (int q = 0; q < LB; ++q) {
          ldadd[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_add, offs[ps + q], 0, 0));
          ldmsk[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_msk, offs[ps + q], 0, 0));
          ldz[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bz, offs[ps + q], 0, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const bool ok = (okmask >> ps) & 1u;
    const unsigned off = offs[ps];
    const float* sp = src + ps * (8 * 4 * BNH);
    const f32x4 ya = *(const f32x4*)sp, yb = *(const f32x4*)(sp + JSTRIDE), yc = *(const f32x4*)(sp + 2 * JSTRIDE);
    f32x4 v = (ya + sgn * yb) + sgn * yc;
    keep[ps] = v;
    if (ok) fs += v;
    if (has_affine) v = v * sc + sh;
    if constexpr (LOADS) {
      v += ldadd[ps % LB];
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = (has_mask && !(ldmsk[ps % LB][c] > 0.f)) ? 0.f : v[c];
    }
    if (do_relu) {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
    }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rs_dst, off, 0, 0);
    if (LOADS && want_bwd) {
      const f32x4 zz = ldz[ps % LB];
      if (ok) {
        s1 += v;
        s2 += v * ((zz - bmu) * bis);
      }
    }
  }
  W2_STAMP(5 + 3 * ((n0h >> 5) & 1));
  if (want_bwd) {
    s1 = wave_rows_sum(s1);
    s2 = wave_rows_sum(s2);
    if (elane < CPR) {
      *(f32x4*)&red1[ewave * BNH + cq * 4] = s1;
      *(f32x4*)&red2[ewave * BNH + cq * 4] = s2;
    }
    __syncthreads();
    if (tid < BNH) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { t1 += red1[w * BNH + tid]; t2 += red2[w * BNH + tid]; }
      const int nn = n0h + tid;
      if (nn < p.N) {
        p.bwd_partial[(mblk * 2 + 0) * p.N + nn] = t1;
        p.bwd_partial[(mblk * 2 + 1) * p.N + nn] = t2;
      }
    }
  }
#if SLIC_W2_ABL & 2048
  if (false) {      // diagnostic: the forward's statistics skipped although asked for (scripts/r5/ab_epilogue.sh)
#else
  if (want_stats) {
#endif
    // BatchNorm partials over this block's real outputs, per channel: (sum v, sum (v - mean_blk)^2), the second from the kept values
    const int64_t left = p.M - mblk * (int64_t)full_rows;
    const float inv_rows = 1.0f / (float)(left < full_rows ? left : full_rows);
    if (want_bwd) __syncthreads();
    fs = wave_rows_sum(fs);
    if (elane < CPR) *(f32x4*)&red1[ewave * BNH + cq * 4] = fs;
    __syncthreads();
    if (tid < BNH) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red1[w * BNH + tid];
      bmean[tid] = t * inv_rows;
      const int nn = n0h + tid;
      if (nn < p.N) p.stat_partial[(mblk * 2 + 0) * p.N + nn] = t;
    }
#if SLIC_W2_ABL & 1024
    return;          // diagnostic: the sums only, no second moment
#endif
    __syncthreads();
    const f32x4 mu = *(const f32x4*)&bmean[cq * 4];
    f32x4 q2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      if ((okmask >> ps) & 1u) {
        const f32x4 d = keep[ps] - mu;
        q2 += d * d;
      }
    }
    q2 = wave_rows_sum(q2);
    if (elane < CPR) *(f32x4*)&red2[ewave * BNH + cq * 4] = q2;
    __syncthreads();
    if (tid < BNH) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red2[w * BNH + tid];
      const int nn = n0h + tid;
      if (nn < p.N) p.stat_partial[(mblk * 2 + 1) * p.N + nn] = t;
    }
  }
}

__device__ __forceinline__ void w2_epilogue(const SlicConvArgs& p, float* lds, int64_t tile0, int n0h, int tid, int full_rows) {
  if (p.addend || p.mask_src || p.bwd_z) w2_epilogue_impl<true>(p, lds, tile0, n0h, tid, full_rows);      // workgroup-uniform
  else w2_epilogue_impl<false>(p, lds, tile0, n0h, tid, full_rows);
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_wino2_kernel(const SlicConvArgs p, const int full_rows, float* __restrict__ slab, const int mb_off, const int mb_cnt) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = wave & 3, th = wave >> 2;
  const int r = lane & 31, hh = lane >> 5;
  const int bx = blockIdx.x, gdx = gridDim.x;
  int mb = (bx & 7) * (gdx >> 3) + (bx >> 3);                // XCD-aware order: neighbouring tile blocks share an L2
  int nb = blockIdx.y, zpiece = blockIdx.z;
#ifndef SLIC_W2_UMAP
#define SLIC_W2_UMAP 1
#endif
#if SLIC_W2_UMAP
  // Round 6 — launches of at most EIGHT tile blocks (layer4 at B = 32: 512 tiles, 8 n blocks x 4 K pieces): there the U operand is the traffic
  // (75 MB at layer4, a different 2.3 MB slice per (n block, K piece)), and with the order above every XCD runs ONE tile block against ALL
  // 32 slices — each XCD streams the whole U through its 4 MB L2, 600 MB per launch.  Turned round: XCD x takes the (n block, K piece)
  // combinations x, x + 8, ... and runs all eight tile blocks against each — a slice is read by one XCD only and shared by eight workgroups
  // through its L2.  (The hardware deals workgroups to XCDs round-robin in dispatch order, x fastest: with gridDim.x == 8, XCD = blockIdx.x.)
  if (gdx == 8 && ((gridDim.y * gridDim.z) & 7) == 0) {
    const int q = (int)(blockIdx.y + gridDim.y * blockIdx.z);
    const int combo = bx + 8 * (q >> 3);
    mb = q & 7;
    nb = combo % (int)gridDim.y;
    zpiece = combo / (int)gridDim.y;
  }
#endif
  const int C = p.Cs, T = p.Ts, H = p.Hs, W = p.Ws;
  const int Wq = (W + 3) >> 2, Hq = (H + 1) >> 1;
  const int64_t Mt = (p.M / ((int64_t)H * W)) * Hq * Wq;     // tiles
  // this launch covers tile blocks [mb_off, mb_off + mb_cnt): all of them, or — a launch whose last dispatch round would be partly
  // filled, or one of few workgroups — the whole rounds with the K loop in one piece, and then the remaining blocks with the K loop
  // cut into gridDim.z even pieces: workgroup z = blockIdx.z of the second launch reduces double stages [z, z + 1) * (3 C / 8) / gridDim.z
  // of the same (channel group, kt) order and writes its two output rows' partial sums to
  // slab[block][piece][column half][row][tile][col][n 32]; conv_wino2_finish adds the pieces in order and runs the epilogue.
  // One piece (slab == NULL) covers everything.
  if (mb >= mb_cnt) return;
  const int64_t tile0 = (int64_t)(mb_off + mb) * 64;
  if (tile0 >= Mt) return;
  const int n0 = nb * 64;
  W2_STAMP_ID();
  W2_STAMP(1);
  __builtin_amdgcn_s_setprio(SLIC_PRIO_EDGE);
  const int CCH = C >> 2;                                     // 4-channel stages per kt (a power of two >= 16: checked on the host)
  const int NB = p.N >> 6;
  // ---- DMA roles.  Pixels: a DOUBLE stage (8 channels) at a time — piece pc = 8 i + wave (i = 0..5) is patch pixel ab = pc / 2 of
  // tile half pc % 2; lane L serves tile 32 (pc % 2) + L % 32, 16-byte half L / 32 of the pixel's 8 channels: lanes L and L + 32 fetch
  // the two halves of ONE 32-byte piece of a cache line (16-byte pieces from 64 different lines per instruction cost 8 % of the
  // kernel), and the LDS image [ab][tile half][channel half][tile 32][4 ch] hands each stage of the pair its own conflict pattern-free
  // half.  A thread thus serves ONE tile (wave % 2 picks the half) at six patch positions ab = 4 i + wave / 2.
  const int64_t mytile = tile0 + (wave & 1) * 32 + (lane & 31);
  const bool tvalid = mytile < Mt;
  unsigned q = (unsigned)(tvalid ? mytile : 0);
  const int wt = (int)(q % (unsigned)Wq); q /= (unsigned)Wq;
  const int h2 = (int)(q % (unsigned)Hq); q /= (unsigned)Hq;  // q = frame (b, t)
  const int tt = (int)(q % (unsigned)T);
  // Stage order: the stage index sl (0 .. 3 C/4 - 1; a K-split piece starts at double stage dbeg and runs 3 C/4 / pieces of them) runs with kt
  // INNER — double stage d = sl / 2 is (channel group cd = d / 3, kt = d % 3), stage sl its channel half sl % 2: a workgroup touches
  // the three frames t - 1, t, t + 1 of its patch in consecutive double stages, and the per-lane piece offsets never change (kt rides in
  // the scalar offset).  Against a kt-outer order (offsets rewritten three times per workgroup) it measured equal in time; FETCH_SIZE
  // per launch 376 vs 334 MB at layer1, 88 vs 107 MB at layer3 (scripts/r4/pmc_fetch_ab.sh).
  // Per-thread state: the six piece offsets (the pixel's address in frame t, or an out-of-range offset where the pixel is outside the
  // frame) wait in LDS behind the rings (LDS instructions cost the matrix pipe nothing; six registers held through the loop would
  // spill); whether frame t - 1 / t + 1 exists is one flag word per lane.  A double stage then costs ONE vector instruction + one per
  // piece: kt and the channel group ride in the DMA's SCALAR offset (the resource starts one frame before the tensor, so that the
  // scalar part is never negative); U = a per-lane constant + the stage's block as scalar offset.
#if SLIC_W2_ABL & 16384
  // diagnostic (scripts/r6/ab_taxis.sh): the K loop of two of the three kt — whole-K launches only (a K-split piece keeps its range: its first
  // double stage rides in the DMA's scalar offset, which no range check sees; the first version of this build cut the pieces too and faulted)
  const int NSL = slab ? 3 * CCH / (int)gridDim.z : 2 * CCH;
#else
  const int NSL = slab ? 3 * CCH / (int)gridDim.z : 3 * CCH;   // stages of this workgroup (a multiple of 4: checked on the host)
#endif
  const int dbeg = slab ? zpiece * (NSL >> 1) : 0;             // its first double stage
  const unsigned HWC4 = (unsigned)(H * W * C * 4);
  unsigned* stash = (unsigned*)(lds + W2_RING_FLOATS) + tid;   // word i of thread tid at [i][tid]: conflict-free (a row per thread was 8-way)
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int ab = 4 * i + (wave >> 1);
    const int a = (ab * 11) >> 6, b = ab - 6 * a;             // ab / 6 for ab < 24
    const int hr = 2 * h2 - 1 + a, wc = 4 * wt - 1 + b;
    const bool ok = tvalid && (unsigned)hr < (unsigned)H && (unsigned)wc < (unsigned)W;
    stash[i * 512] = ok ? (unsigned)(((((int64_t)q * H + hr) * W + wc) * C) * 4) + (unsigned)(lane >> 5) * 16u : 0xFFFFFF00u;   // past any resource wino2_check admits (a raw buffer checks the VECTOR offset only)
  }
  // bit kt: frame t - 1 + kt is outside the clip; bit 3: a dead stage
  const int tflags = (tt == 0 ? 1 : 0) | (tt == T - 1 ? 4 : 0) | 8;
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.src - HWC4), 0, p.src_bytes + 2 * HWC4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wgt = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, 0, p.wgt_bytes, 0x00020000);
  const unsigned uvoff = (unsigned)tid * 16u;
  // (kt, channel group) of local double stage d
  auto kt_cd = [&](int d, int& kt, int& cd) {
    d += dbeg;
    cd = (int)(((unsigned)d * 43691u) >> 17);                  // d / 3 for d < 2^16
    kt = d - 3 * cd;
  };
  // pixel double stage d into pixel slot `slot`, in two parts of three pieces (three offset registers live at a time)
  auto issue_px = [&](int d, int slot, int part) {
    const bool live = 2 * d < NSL;
    int kt, cd;
    kt_cd(live ? d : 0, kt, cd);
    const unsigned inv = (unsigned)__builtin_amdgcn_sbfe(tflags, live ? kt : 3, 1);          // -1: this frame does not exist
    const unsigned soff = (unsigned)kt * HWC4 + (unsigned)(cd * 32);      // 8 channels = 32 bytes per double stage
#if SLIC_W2_ABL & 512
    {
      // diagnostic: 32 contiguous 1 KB runs per double stage and workgroup (wave w: runs 4 w .. 4 w + 3 = image rows 2 w, 2 w + 1 of the block's
      // 16, two halves each), addressed as the planar layout would be: row pitch W C 4 bytes, channel-group pitch W 32 bytes
      const int rows_total = (int)(p.M / W);
      const int row0 = (int)(tile0 / (unsigned)Wq) * 2 - 1;      // first image row of the block's patch rows (all frames stacked)
#pragma unroll
      for (int u = 2 * part; u < 2 * part + 2; ++u) {
        const int id = 4 * wave + u;
        // the row of frame t - 1 + kt, clamped INTO the tensor (the scalar offset is not range-checked; a block may straddle two frames, so the
        // lanes' own frame flags do not cover every row fetched here): [0, rows_total - 2] leaves room for the run's 2 KB
        int rg = row0 + (id >> 1) + (kt - 1) * H;
        rg = rg < 0 ? 0 : (rg > rows_total - 2 ? rows_total - 2 : rg);
        const unsigned so2 = HWC4 + (unsigned)rg * (unsigned)(W * C * 4) + (unsigned)(cd * W * 32) + (unsigned)((id & 1) * 1024);
        const unsigned off = ((unsigned)lane * 16u) | inv;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void*)(lds + slot * W2_PX_FLOATS + id * 256),
                                                 16, (int)off, (int)so2, 0, SLIC_W2_PAUX);
      }
      return;
    }
#endif
#pragma unroll
    for (int u = 3 * part; u < 3 * part + 3; ++u) {
      constexpr int ORD[6] = SLIC_W2_PXORDER;
      const int i = ORD[u];
#if SLIC_W2_ABL & 1
      const unsigned off = 0xFFFFFF00u + 0 * (stash[i * 512] | inv);
#else
      const unsigned off = stash[i * 512] | inv;
#endif
      // piece pc = 8 i + wave = (ab, tile half): image [ab 24][slot 2][tile half 2][channel half 2][tile 32][4 ch] — the two ring slots
      // are INTERLEAVED per patch pixel, so that one lane address per patch row reaches both slots with immediate offsets
      const int pc = 8 * i + wave;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void*)(lds + (((pc >> 1) * 2 + slot) * 2 + (pc & 1)) * 256),
                                               16, (int)off, (int)soff, 0, SLIC_W2_PAUX);
#if SLIC_W2_ABL & 32768
      // diagnostic: every piece fetched twice — a neighbouring frame's pixels into the same place (twice the L2 -> LDS bytes per MFMA)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void*)(lds + (((pc >> 1) * 2 + slot) * 2 + (pc & 1)) * 256),
                                               16, (int)off, (int)(soff ^ 64u), 0, SLIC_W2_PAUX);
#endif
    }
  };
  // U block of local stage sl into U slot `slot`
  auto issue_u = [&](int sl, int slot) {
    const bool live = sl < NSL;
    int kt, cd;
    kt_cd(live ? (sl >> 1) : 0, kt, cd);
    const int cc = 2 * cd + (live ? (sl & 1) : 0);
    const unsigned ublk = (unsigned)((kt * CCH + cc) * NB + nb) * (unsigned)(W2_U_FLOATS * 4);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#if SLIC_W2_ABL & 1
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (__attribute__((address_space(3))) void*)(lds + 2 * W2_PX_FLOATS + slot * W2_U_FLOATS + (i * 512 + wave * 64) * 4),
                                               16, (int)(0xFFFFFF00u + 0 * uvoff), (int)(0 * ublk), 0, 0);
#else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (__attribute__((address_space(3))) void*)(lds + 2 * W2_PX_FLOATS + slot * W2_U_FLOATS + (i * 512 + wave * 64) * 4),
                                               16, (int)uvoff, (int)(ublk + (unsigned)(i * 8192)), 0, SLIC_W2_UAUX);
#if SLIC_W2_ABL & 32768
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (__attribute__((address_space(3))) void*)(lds + 2 * W2_PX_FLOATS + slot * W2_U_FLOATS + (i * 512 + wave * 64) * 4),
                                               16, (int)uvoff, (int)((ublk + (unsigned)(i * 8192)) ^ 32768u), 0, SLIC_W2_UAUX);
#endif
#endif
    }
  };
  f32x16 acc[6][2];
#pragma unroll
  for (int pp = 0; pp < 6; ++pp)
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[pp][nh][g] = 0.f;
  // prologue: pixel double stage 0, U of stage 0 — in the order of the steady state (per stage U first, then pixels)
  issue_u(0, 0);
  issue_px(0, 0, 0);
  issue_px(0, 0, 1);
  W2_STAMP(2);
  __builtin_amdgcn_s_setprio(0);
  // reader offsets (floats): pixel (a, b) of this lane's tile, channel half e2 of the double stage, its channel pair:
  //   (((ab * 2 + slot) * 2 + th) * 2 + e2) * 128 + r * 4 + 2 hh;   point (j, p) of column half nh: ((j * 6 + p) * 2 + nh) * 128 + hh * 64 + r * 2
  // (U is stored channel pair major — the 32 lanes of a ds_read_b64's lane group then cover one 256-byte bank row; the pixel image's
  //  order is the DMA's, 16 bytes per tile: lanes r and r + 16 of a group share banks, two LDS cycles per group instead of one)
  const int a1 = j == 0 ? 0 : 1, a2 = j == 3 ? 3 : 2;
  // THREE lane addresses (bytes) serve every LDS read of the loop with immediate offsets; they are made opaque to the compiler, which
  // otherwise re-associates the large constants into one address register per read and spills them (a scratch reload in the loop
  // also drains the DMA queue: vmcnt counts it)
  typedef __attribute__((address_space(3))) const char* lds_cptr;
  const unsigned lbase = (unsigned)(size_t)((__attribute__((address_space(3))) float*)lds);
  unsigned ar1 = lbase + (unsigned)(a1 * 6144 + th * 256 + r * 4 + 2 * hh) * 4u;      // + (b * 1024 + slot * 512 + e2 * 128) * 4
  unsigned ar2 = lbase + (unsigned)(a2 * 6144 + th * 256 + r * 4 + 2 * hh) * 4u;
  unsigned bro = lbase + (unsigned)(2 * W2_PX_FLOATS + j * 6 * 256 + hh * 64 + r * 2) * 4u;   // + (slot * W2_U_FLOATS + p * 256 + nh * 128) * 4
  asm volatile("" : "+v"(ar1), "+v"(ar2), "+v"(bro));
  const float sg = j == 1 ? 1.f : -1.f;                       // H-point: d[a1] + sg * d[a2]  (j = 2 as d1 - d2: its U is negated)
  const f32x2 sgn = {sg, sg};
  const f32x2 c2 = {2.f, 2.f}, c4 = {4.f, 4.f}, c5 = {5.f, 5.f};
  f32x2 V[6], ut[2];                                          // ut: U of point 5 (kept across the stage barrier with V[5])
#pragma unroll
  for (int pp = 0; pp < 6; ++pp) V[pp] = (f32x2){0.f, 0.f};
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) ut[nh] = (f32x2){0.f, 0.f};
  // the four MFMAs of point pp: element e of the lane's channel pair goes to MFMA e (the two column halves alternate: the same
  // accumulator comes round every other MFMA, 128 cycles apart)
  auto mfma_point = [&](const int pp, const f32x2 (&u)[2]) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) acc[pp][nh] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[pp][e], u[nh][e], acc[pp][nh], 0, 0, 0);
  };
  auto read_u = [&](const int uslot, const int pp, f32x2 (&u)[2]) {
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) u[nh] = *(const __attribute__((address_space(3))) f32x2*)((lds_cptr)bro + (uslot * W2_U_FLOATS + pp * 256 + nh * 128) * 4);
  };
  // Ring discipline.  Pixel slots: 2 x 48 KB, double stage ds in slot ds % 2, issued WHOLE (six pieces) in stage 2 ds - 2, i.e.
  // two stages ahead, behind the barrier that ends the last reads of double stage ds - 2.  U slots: 2 x 24 KB, stage s in slot s % 2,
  // issued first thing in stage s - 1.  vmcnt retires in order, so the wait in front of an odd stage leaves the six pixel pieces
  // in flight (vmcnt(6)) and the one in front of an even stage takes everything (vmcnt(0): the pixels issued two stages ago, the U
  // issued one stage ago).  Four stage bodies per loop turn: (double-stage parity, channel half) fix every LDS slot at compile time.
#if SLIC_W2_ABL & 512
  constexpr unsigned WAIT_VM6_LGKM0 = 4 | 0x70, WAIT_VM0_LGKM0 = 0x70;      // four pixel runs per wave and double stage
#elif SLIC_W2_ABL & 32768
  constexpr unsigned WAIT_VM6_LGKM0 = 12 | 0x70, WAIT_VM0_LGKM0 = 0x70;     // twelve pixel pieces per wave and double stage
#else
  constexpr unsigned WAIT_VM6_LGKM0 = 6 | 0x70, WAIT_VM0_LGKM0 = 0x70;
#endif
  for (int s0 = 0; s0 < NSL; s0 += 4) {
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
      const int sgl = s0 + sidx;
      const int e2 = sidx & 1, pslot = (sidx >> 1) & 1, uslot = sidx & 1;
      // (the builtin, not inline assembly: the compiler's own wait-count pass then knows that the registers loaded from LDS in the
      // previous stage — ut — are in, and does not put an lgkmcnt(0) between this stage's reads and the MFMAs that cover them)
      if (e2) __builtin_amdgcn_s_waitcnt(WAIT_VM6_LGKM0);
      else __builtin_amdgcn_s_waitcnt(WAIT_VM0_LGKM0);
#if !(SLIC_W2_ABL & 2)
      __builtin_amdgcn_s_barrier();
#endif
      const int pso = (pslot * 512 + e2 * 128) * 4;            // bytes
      // A: the stage's pixel reads, all issued at once
      f32x2 d1[6], d2[6], cmb[6];
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        d1[b] = *(const __attribute__((address_space(3))) f32x2*)((lds_cptr)ar1 + pso + b * 4096);
        d2[b] = *(const __attribute__((address_space(3))) f32x2*)((lds_cptr)ar2 + pso + b * 4096);
      }
      f32x2 ua[2], ub[2];
      __builtin_amdgcn_sched_barrier(0);
      // B: the next stage's U, then the previous stage's last point (V[5], ut still hold it) under the latency of A
      issue_u(sgl + 1, uslot ^ 1);
      mfma_point(5, ut);
      __builtin_amdgcn_sched_barrier(0);
      read_u(uslot, 0, ua);                                       // lands under the transform
      __builtin_amdgcn_sched_barrier(0);
      // C: H-point, then V = B^T (.) along W.  The packed ops are inline assembly, which the compiler's hazard recogniser does not see
      // as VALU (an MFMA reading a register within two instructions of the op that wrote it would read the OLD value): one fenced
      // block closed by the two wait states.
#pragma unroll
      for (int b = 0; b < 6; ++b) cmb[b] = pk_fma(d2[b], sgn, d1[b]);
      wino_bt6(cmb, V, c2, c4, c5);
      asm volatile("s_nop 1" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // D: the stage's MFMAs point by point, the next point's U fetched one point ahead (two points' fragments live at a time);
      // an even stage issues the pixel double stage two stages ahead between them
      read_u(uslot, 1, ub);
      mfma_point(0, ua);
      read_u(uslot, 2, ua);
      mfma_point(1, ub);
      __builtin_amdgcn_sched_barrier(0);
      if (!e2) issue_px((sgl >> 1) + 1, pslot ^ 1, 0);
      read_u(uslot, 3, ub);
      mfma_point(2, ua);
      read_u(uslot, 4, ua);
      mfma_point(3, ub);
      __builtin_amdgcn_sched_barrier(0);
      if (!e2) issue_px((sgl >> 1) + 1, pslot ^ 1, 1);
      read_u(uslot, 5, ut);
      mfma_point(4, ua);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  mfma_point(5, ut);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  W2_STAMP(3);
  if (tid < 64) w2_tile_records(p, lds, tile0, tid);           // visible behind the barrier in front of w2_epilogue
#if SLIC_W2_ABL & 32
  if (acc[0][0][0] != 12345.678f) return;                      // diagnostic build: no epilogue
#endif
  __builtin_amdgcn_s_setprio(SLIC_PRIO_EDGE);
  // Y = A^T M along W,  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]   (in place: acc[0..3] become the four columns)
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
    const f32x16 s12 = acc[1][nh] + acc[2][nh], d12 = acc[1][nh] - acc[2][nh];
    const f32x16 s34 = acc[3][nh] + acc[4][nh], d34 = acc[3][nh] - acc[4][nh];
    acc[0][nh] = acc[0][nh] + s12 + s34;
    acc[1][nh] = d12 + 2.f * d34;
    acc[2][nh] = s12 + 4.f * s34;
    acc[3][nh] = d12 + 8.f * d34 + acc[5][nh];
  }
  // The four H-points of a tile meet in LDS, one column half at a time: every wave writes its W-outputs to buf[j][tile][o][n 32]
  // (lanes r = consecutive n: 128-byte runs), a barrier, and w2_epilogue combines them row by row
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
    if (nh) __syncthreads();                                   // the first half's readers are done with buf and the reduction rows
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int tl = th * 32 + (g & 3) + 8 * (g >> 2) + 4 * hh;
        lds[((j * 64 + tl) * 4 + o) * 32 + r] = acc[o][nh][g];
      }
    __syncthreads();
    W2_STAMP(4 + 3 * nh);
#if SLIC_W2_ABL & 256
    if (tid < 64) p.dst[(tile0 * 8) * p.ldo + n0 + tid] = lds[tid * 97];      // diagnostic build: no row-major epilogue at all
#else
    if (slab) {
      // K-split piece: the two output rows' partial sums, combined as the epilogue combines them, to the slab (16 bytes per lane)
      float* out = slab + ((((int64_t)mb * gridDim.y + nb) * gridDim.z + zpiece) * 2 + nh) * (2 * 64 * 4 * 32);
      const int cq = tid & 7, rr = tid >> 3;
      const int hp = (rr >> 2) & 1, o = rr & 3;
      const float sgn = hp ? -1.f : 1.f;
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        const int tl = ps * 8 + (rr >> 3);
        const float* sp = lds + hp * (64 * 4 * 32) + (tl * 4 + o) * 32 + cq * 4;
        const f32x4 ya = *(const f32x4*)sp, yb = *(const f32x4*)(sp + 64 * 4 * 32), yc = *(const f32x4*)(sp + 2 * 64 * 4 * 32);
        *(f32x4*)(out + ((hp * 64 + tl) * 4 + o) * 32 + cq * 4) = (ya + sgn * yb) + sgn * yc;
      }
    } else {
      w2_epilogue(p, lds, tile0, n0 + nh * 32, tid, full_rows);
      W2_STAMP(6 + 3 * nh);
    }
#endif
  }
#ifdef SLIC_W2_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // diagnostic build only: this wave's stores have been acknowledged
  W2_STAMP(10);
#endif
}

// ------------------------------------------------------------------------------------------
// Round 6: the PERSISTENT form of conv_wino2_kernel, for launches of many FULL blocks (H even, W % 4 == 0, every block 64 whole tiles,
// one K piece; layers 1 and 2 at B = 32).  In-kernel stamps of the one-block-per-workgroup kernel (scripts/r6/stamps_wino2.py,
// profiles/r06_wino2_stamps.txt; layer1 forward, us per workgroup): decode + first DMAs issued 2.1, K loop 80.8, epilogue 8.1 (image of
// column half 0 written 1.1, its passes 1.5, its statistics 1.6, half 1: 1.0 / 1.3 / 1.4, stores acknowledged 0.2), hand-over to the next
// workgroup of the CU 0.66 — 11 of 91.7 us outside the K loop, none of it hidden: one workgroup fills the CU's LDS.  Here a workgroup
// walks a list of blocks and
//   * the next block's piece offsets are decoded while the waves sit at the epilogue's barriers, and its first U stage and pixel double
//     stage are ISSUED as soon as the last read of the epilogue's LDS image is behind a barrier — they land under the statistics'
//     reduction and the stores; no workgroup launch, no kernel-argument loads, no first-DMA latency per block;
//   * the BatchNorm statistics of BOTH column halves are reduced once, from registers, behind ONE barrier (they were three
//     barrier-separated steps per half): a thread keeps (sum, sum of squares) of its eight values per channel about its FIRST value
//     (a sample of the same distribution: no cancellation to speak of), the row groups of a wave and then the eight waves are merged
//     with Chan's formula in a fixed order — the same quantities (sum v, sum (v - mean_block)^2) as w2_epilogue_impl writes.
// Blocks are dealt statically: XCD x (workgroups with blockIdx.x % 8 == x) owns tile blocks [x gx / 8, (x + 1) gx / 8) of every n block, and
// its workgroups take consecutive entries of that list round-robin — at any moment the workgroups of an XCD work on neighbouring tile
// blocks of one n block (shared halo rows and U block in its L2), as the hardware's dispatch order gives the one-shot kernel.
// lane id WITHOUT the work-item id register: in the persistent kernel every value derived from threadIdx.x that lives across the K loop is a
// register the allocator spills, and a scratch reload in the epilogue waits for vmcnt(0) — i.e. for the next block's DMAs that were just
// issued.  v_mbcnt recomputes the lane where it is needed (volatile: not hoisted out of the block loop); the wave index is scalar.
__device__ __forceinline__ int w2_lane_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

struct W2Stats {
  f32x4 ref[2], sd[2], sq[2], s1[2], s2[2];
};

// The passes of one column half over the LDS image buf[j 4][tile 64][col 4][n 32] (see w2_epilogue_impl), every tile whole, in TWO steps:
//   w2p_read  — the thread's eight output rows combined from the image (row 0 = (Y0 + Y1) + Y2, row 1 = (Y1 - Y2) - Y3) into registers;
//   w2p_write — (ReLU,) store, statistics accumulated into `st` (registers; no reduction here).  No per-channel or per-element operand: a
//               load of one issued behind the next block's DMAs makes its first use wait for those DMAs (one in-order counter) — launches
//               with an affine, an addend, a mask or BatchNorm-backward sums take the LOADS kernel and w2p_passes_loads.
// The second half's steps are split by the barrier that frees the ring: the next block's first DMAs are issued BETWEEN them, in front of this
// half's stores in the vector-memory queue (issued behind the stores they waited ~1 us for the queue to drain: scripts/r6/stamps_wino2p.py).
struct W2Half {
  f32x4 v[8];
  unsigned offs[8];
};

__device__ __forceinline__ void w2p_read(const SlicConvArgs& p, const float* lds, const int n0h, const int ewave, W2Half& h) {
  constexpr int BNH = 32, NPASS = 8;
  constexpr int JSTRIDE = 64 * 4 * BNH;
  const int tid = ewave * 64 + w2_lane_now();
  const int cq = tid & 7, rr = tid >> 3;
  const int n = n0h + cq * 4;
  const int hp = (rr >> 2) & 1, o = rr & 3;
  const float sgn = hp ? -1.f : 1.f;
  const float* src = lds + hp * JSTRIDE + (((rr >> 3) * 4 + o) * BNH + cq * 4);
  const uint2* trec = (const uint2*)(lds + W2_TREC_FLOATS) + (rr >> 3);
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) h.offs[ps] = ((trec[ps * 8].x + (unsigned)(hp * p.Ws + o)) * (unsigned)p.ldo + (unsigned)n) * 4u;
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const float* sp = src + ps * (8 * 4 * BNH);
    const f32x4 ya = *(const f32x4*)sp, yb = *(const f32x4*)(sp + JSTRIDE), yc = *(const f32x4*)(sp + 2 * JSTRIDE);
    h.v[ps] = (ya + sgn * yb) + sgn * yc;
  }
}

template <int NH>
__device__ __forceinline__ void w2p_write(const SlicConvArgs& p, W2Half& h, W2Stats& st) {
  constexpr int NPASS = 8;
  const bool want_stats = p.stat_partial != nullptr, do_relu = p.relu != 0;
  const unsigned dst_bytes = (unsigned)(((p.M - 1) * (int64_t)p.ldo + p.N) * 4);
  const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc((void*)p.dst, 0, dst_bytes, 0x00020000);
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    f32x4 v = h.v[ps];
    if (want_stats) {
      if (ps == 0) st.ref[NH] = v;
      const f32x4 d = v - st.ref[NH];
      st.sd[NH] += d;
      st.sq[NH] += d * d;
    }
    if (do_relu) {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
    }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rs_dst, h.offs[ps], 0, 0);
  }
}

// the same in ONE step for the launches that load optional operands (addend / ReLU-backward mask / the BatchNorm z of a data gradient): their
// loads are issued four passes at a time, and holding eight combined rows beside them spilled registers
#ifndef SLIC_W2P_LB0
#define SLIC_W2P_LB0 4     // passes whose optional-operand loads are in flight together, column half 0 / 1 (8 = all of a half's at once)
#endif
#ifndef SLIC_W2P_LB1
#define SLIC_W2P_LB1 8
#endif
template <int NH>
__device__ __forceinline__ void w2p_passes_loads(const SlicConvArgs& p, const float* lds, const int n0h, const int ewave, W2Stats& st) {
  constexpr int BNH = 32, NPASS = 8;
  const int tid = ewave * 64 + w2_lane_now();
  constexpr int JSTRIDE = 64 * 4 * BNH;
  const int W = p.Ws;
  const bool want_stats = p.stat_partial != nullptr, want_bwd = p.bwd_partial != nullptr;
  const bool has_mask = p.mask_src != nullptr, do_relu = p.relu != 0;
  const bool has_affine = p.scale != nullptr || p.shift != nullptr;
  const unsigned dst_bytes = (unsigned)(((p.M - 1) * (int64_t)p.ldo + p.N) * 4);
  const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc((void*)p.dst, 0, dst_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc((void*)p.addend, 0, p.addend ? dst_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_msk = __builtin_amdgcn_make_buffer_rsrc((void*)p.mask_src, 0, has_mask ? dst_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bz = __builtin_amdgcn_make_buffer_rsrc((void*)p.bwd_z, 0, p.bwd_z ? dst_bytes : 0, 0x00020000);
  const int cq = tid & 7, rr = tid >> 3;
  const int n = n0h + cq * 4;
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f}, bmu = sh, bis = sh;
  if (p.scale) sc = *(const f32x4*)(p.scale + n);
  if (p.shift) sh = *(const f32x4*)(p.shift + n);
  if (want_bwd) { bmu = *(const f32x4*)(p.bwd_mean + n); bis = *(const f32x4*)(p.bwd_invstd + n); }
  const int hp = (rr >> 2) & 1, o = rr & 3;
  const float sgn = hp ? -1.f : 1.f;
  const float* src = lds + hp * JSTRIDE + (((rr >> 3) * 4 + o) * BNH + cq * 4);
  const uint2* trec = (const uint2*)(lds + W2_TREC_FLOATS) + (rr >> 3);
  unsigned offs[NPASS];
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) offs[ps] = ((trec[ps * 8].x + (unsigned)(hp * W + o)) * (unsigned)p.ldo + (unsigned)n) * 4u;
  // (the second half's accumulators are dead by then: all eight passes' loads fit the registers and pay ONE memory round trip — these
  //  tensors come from HBM, ~2 us — where batches of four paid two; the first half holds the second's accumulators beside them)
  constexpr int LB = NH ? SLIC_W2P_LB1 : SLIC_W2P_LB0;
  f32x4 ldadd[LB], ldmsk[LB], ldz[LB];
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    if (ps % LB == 0) {
#pragma unroll
      for (int q = 0; q < LB; ++q) {
        ldadd[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_add, offs[ps + q], 0, 0));
        ldmsk[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_msk, offs[ps + q], 0, 0));
        ldz[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bz, offs[ps + q], 0, 0));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const float* sp = src + ps * (8 * 4 * BNH);
    const f32x4 ya = *(const f32x4*)sp, yb = *(const f32x4*)(sp + JSTRIDE), yc = *(const f32x4*)(sp + 2 * JSTRIDE);
    f32x4 v = (ya + sgn * yb) + sgn * yc;
    if (want_stats) {
      if (ps == 0) st.ref[NH] = v;
      const f32x4 d = v - st.ref[NH];
      st.sd[NH] += d;
      st.sq[NH] += d * d;
    }
    if (has_affine) v = v * sc + sh;
    v += ldadd[ps % LB];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = (has_mask && !(ldmsk[ps % LB][c] > 0.f)) ? 0.f : v[c];
    if (do_relu) {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
    }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rs_dst, offs[ps], 0, 0);
    if (want_bwd) {
      const f32x4 zz = ldz[ps % LB];
      st.s1[NH] += v;
      st.s2[NH] += v * ((zz - bmu) * bis);
    }
  }
}

__device__ __forceinline__ float* w2p_red() {
  __shared__ float red[8 * 2 * 64];
  return red;
}

// The value of lanes L and L ^ 8 / L ^ 16 / L ^ 32 as a pair (lo, hi) WITHOUT the LDS crossbar: a DPP row rotation (rows of 16 lanes) and
// gfx950's v_permlane16_swap / v_permlane32_swap — vector instructions of a few cycles where ds_bpermute_b32 is an LDS round trip per step
// (the three-level butterflies of the statistics measured 1.2 us per block with it).  Every use below is symmetric in the pair.
// (inline assembly for the swaps: through __builtin_amdgcn_permlane16_swap(x, x) this compiler used the FIRST result for both halves of the pair
//  — `v_sub_f32 v28, v22, v22` behind `v_permlane16_swap_b32 v22, v28` — and every statistic came out wrong; the test caught it.  The two
//  wait states cover the VALU write of the operands in front of the swap, as the compiler's own sequence does.)
__device__ __forceinline__ void w2_pair(const float x, const int off, float& lo, float& hi) {
  if (off == 8) {
    lo = x;
    hi = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));      // row_ror:8
  } else if (off == 16) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    lo = a;
    hi = b;
  } else {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    lo = a;
    hi = b;
  }
}

// one reduction for both column halves, behind one barrier (the caller has passed the barrier that ends the image reads); first half: the row
// groups of a wave merged by lane exchanges, the wave's rows to LDS
template <int NHN>
__device__ __forceinline__ void w2p_stats_finish(const SlicConvArgs& p, float* lds, const int64_t mblk, const int n0, const int ewave, W2Stats& st) {
  const bool want_stats = p.stat_partial != nullptr, want_bwd = p.bwd_partial != nullptr;
  if (!want_stats && !want_bwd) return;                          // workgroup-uniform
  const bool do_bwd = want_bwd && !want_stats;                   // (never both: see w2p_red)
  float* red = w2p_red();
  const int elane = w2_lane_now(), cq = elane & 7;
#pragma unroll
  for (int nh = 0; nh < NHN; ++nh) {
    if (want_stats) {
      f32x4 sum = 8.f * st.ref[nh] + st.sd[nh];
      f32x4 m2 = st.sq[nh] - st.sd[nh] * st.sd[nh] * 0.125f;
      // Chan, equal counts n on both sides: M2 = M2a + M2b + (sa - sb)^2 / (2 n); n = 8, 16, 32
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float sv = sum[c], mv = m2[c];
#pragma unroll
        for (int lv = 0; lv < 3; ++lv) {
          float sa, sb, ma, mb;
          w2_pair(sv, 8 << lv, sa, sb);
          w2_pair(mv, 8 << lv, ma, mb);
          const float dm = sa - sb;
          mv = (ma + mb) + dm * dm * (1.f / (float)(16 << lv));
          sv = sa + sb;
        }
        sum[c] = sv;
        m2[c] = mv;
      }
      if (elane < 8) {
        *(f32x4*)&red[(ewave * 2 + 0) * 64 + nh * 32 + cq * 4] = sum;
        *(f32x4*)&red[(ewave * 2 + 1) * 64 + nh * 32 + cq * 4] = m2;
      }
    }
    if (do_bwd) {
      f32x4 a = st.s1[nh], b = st.s2[nh];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int lv = 0; lv < 3; ++lv) {
          float lo, hi;
          w2_pair(a[c], 8 << lv, lo, hi);
          a[c] = lo + hi;
          w2_pair(b[c], 8 << lv, lo, hi);
          b[c] = lo + hi;
        }
      if (elane < 8) {
        *(f32x4*)&red[(ewave * 2 + 0) * 64 + nh * 32 + cq * 4] = a;
        *(f32x4*)&red[(ewave * 2 + 1) * 64 + nh * 32 + cq * 4] = b;
      }
    }
  }
}

// second half: the eight waves' rows merged by 64 threads.  Called AFTER the next block's first DMAs have been issued (reads of a distinct
// static array: the compiler's wait-count pass does not hold them back behind the in-flight LDS-DMAs; the first half, whose LDS accesses it
// could not tell apart from the DMAs' destinations, drew an s_waitcnt vmcnt when it ran behind the issue)
template <int NHN>
__device__ __forceinline__ void w2p_stats_merge(const SlicConvArgs& p, const int64_t mblk, const int n0, const int ewave) {
  const bool want_stats = p.stat_partial != nullptr, want_bwd = p.bwd_partial != nullptr;
  if (!want_stats && !want_bwd) return;                          // workgroup-uniform
  const bool do_bwd = want_bwd && !want_stats;
  float* red = w2p_red();
  const int tid = ewave * 64 + w2_lane_now();
  // (a raw barrier: __syncthreads() is also a fence, and with the next block's LDS-DMAs in flight the compiler makes it wait for vmcnt(0))
  __builtin_amdgcn_s_waitcnt(0xC07F);                            // lgkmcnt(0): this wave's rows are in LDS
  __builtin_amdgcn_s_barrier();
  if (tid < 32 * NHN) {                                            // (a half item: its 32 columns, n0 = the first of them)
    const int n = n0 + tid;
    float v0[8], v1[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) { v0[w] = red[(w * 2 + 0) * 64 + tid]; v1[w] = red[(w * 2 + 1) * 64 + tid]; }
    if (want_stats) {
      // the eight waves (64 rows each) in order: running (64 w rows: S, M) + (64 rows: sw, mw); d = mean_w - mean_run, weight 64 w * 64 / (64 w + 64)
      float S = v0[0], M = v1[0];
#pragma unroll
      for (int w = 1; w < 8; ++w) {
        const float d = v0[w] * (1.f / 64.f) - S * (1.f / (64.f * (float)w));
        M = (M + v1[w]) + d * d * (64.f * (float)w / (float)(w + 1));
        S += v0[w];
      }
      p.stat_partial[(mblk * 2 + 0) * p.N + n] = S;
      p.stat_partial[(mblk * 2 + 1) * p.N + n] = M;
    }
    if (do_bwd) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { t1 += v0[w]; t2 += v1[w]; }
      p.bwd_partial[(mblk * 2 + 0) * p.N + n] = t1;
      p.bwd_partial[(mblk * 2 + 1) * p.N + n] = t2;
    }
  }
}

// LOADS: the launch reads optional operands in its epilogue (a per-channel affine, an addend, the ReLU-backward mask, the BatchNorm z of a data gradient) — two kernels,
// so that each has one straight-line epilogue (with both behind workgroup-uniform branches the register allocator spilled the rows the
// forward holds across the DMA issue)
// CT: the reduction channel count as a compile-time constant (64 / 128 / 256: the K loop's trip count and the stage -> (kt, channel group)
// arithmetic fold; 0 = read it from the arguments) — which also puts the layer into the kernel's NAME: the profiles tell layer1's launches
// (conv_wino2p_kernel<false, 64>) from layer2's by it, where the one-block kernel's grid size did
template <bool LOADS, int CT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_wino2p_kernel(const SlicConvArgs p, const int gx, const int ny) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = wave & 3, th = wave >> 2;
  const int r = lane & 31, hh = lane >> 5;
  const int C = CT ? CT : p.Cs, T = p.Ts, H = p.Hs, W = p.Ws;
  const int Wq = W >> 2, Hq = H >> 1;
  // ---- this workgroup's list: entries e = l, l + GL, ... of XCD x's list (n block major, its tile blocks inside)
  const int x = blockIdx.x & 7, l = blockIdx.x >> 3, GL = gridDim.x >> 3;
  const int per = (gx + 7) >> 3;                               // tile blocks per XCD
  const int mb_lo = x * per, mb_n = min(per, gx - mb_lo);      // (mb_n <= 0: nothing for this XCD)
  const int nent = mb_n > 0 ? mb_n * ny : 0;
  if (l >= nent) return;
  __builtin_amdgcn_s_setprio(SLIC_PRIO_EDGE);
  const int CCH = C >> 2;
  const int NB = p.N >> 6;
  const int NSL = 3 * CCH;
  const unsigned HWC4 = (unsigned)(H * W * C * 4);
  unsigned* stash = (unsigned*)(lds + W2_RING_FLOATS) + tid;
  // entry -> (first tile, n block); the piece offsets of its DMAs into the stash, its frame flags returned
  auto entry_mb = [&](int e) { return mb_lo + e % mb_n; };
  auto entry_nb = [&](int e) { return e / mb_n; };
  auto decode = [&](int e) -> int {
    const int64_t tile0 = (int64_t)entry_mb(e) * 64;
    unsigned q = (unsigned)(tile0 + (wave & 1) * 32 + (lane & 31));
    const int wt = (int)(q % (unsigned)Wq); q /= (unsigned)Wq;
    const int h2 = (int)(q % (unsigned)Hq); q /= (unsigned)Hq;
    const int tt = (int)(q % (unsigned)T);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int ab = 4 * i + (wave >> 1);
      const int a = (ab * 11) >> 6, b = ab - 6 * a;
      const int hr = 2 * h2 - 1 + a, wc = 4 * wt - 1 + b;
      const bool ok = (unsigned)hr < (unsigned)H && (unsigned)wc < (unsigned)W;
      stash[i * 512] = ok ? (unsigned)(((((int64_t)q * H + hr) * W + wc) * C) * 4) + (unsigned)(lane >> 5) * 16u : 0xFFFFFF00u;
    }
    return (tt == 0 ? 1 : 0) | (tt == T - 1 ? 4 : 0) | 8;
  };
  int tflags = 0, nb_dma = 0;                                   // state of the DMA issue: the entry whose stages are being fetched
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.src - HWC4), 0, p.src_bytes + 2 * HWC4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wgt = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, 0, p.wgt_bytes, 0x00020000);
  const unsigned uvoff = (unsigned)tid * 16u;
  auto kt_cd = [&](int d, int& kt, int& cd) {
    cd = (int)(((unsigned)d * 43691u) >> 17);
    kt = d - 3 * cd;
  };
  auto issue_px = [&](int d, int slot, int part) {
    const bool live = 2 * d < NSL;
    int kt, cd;
    kt_cd(live ? d : 0, kt, cd);
    const unsigned inv = (unsigned)__builtin_amdgcn_sbfe(tflags, live ? kt : 3, 1);
    const unsigned soff = (unsigned)kt * HWC4 + (unsigned)(cd * 32);
#pragma unroll
    for (int u = 3 * part; u < 3 * part + 3; ++u) {
      constexpr int ORD[6] = SLIC_W2_PXORDER;
      const int i = ORD[u];
      const unsigned off = stash[i * 512] | inv;
      const int pc = 8 * i + wave;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void*)(lds + (((pc >> 1) * 2 + slot) * 2 + (pc & 1)) * 256),
                                               16, (int)off, (int)soff, 0, SLIC_W2_PAUX);
    }
  };
  auto issue_u = [&](int sl, int slot) {
    const bool live = sl < NSL;
    int kt, cd;
    kt_cd(live ? (sl >> 1) : 0, kt, cd);
    const int cc = 2 * cd + (live ? (sl & 1) : 0);
    const unsigned ublk = (unsigned)((kt * CCH + cc) * NB + nb_dma) * (unsigned)(W2_U_FLOATS * 4);
#pragma unroll
    for (int i = 0; i < 3; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (__attribute__((address_space(3))) void*)(lds + 2 * W2_PX_FLOATS + slot * W2_U_FLOATS + (i * 512 + wave * 64) * 4),
                                               16, (int)uvoff, (int)(ublk + (unsigned)(i * 8192)), 0, SLIC_W2_UAUX);
  };
  // reader state (as conv_wino2_kernel)
  const int a1 = j == 0 ? 0 : 1, a2 = j == 3 ? 3 : 2;
  typedef __attribute__((address_space(3))) const char* lds_cptr;
  const unsigned lbase = (unsigned)(size_t)((__attribute__((address_space(3))) float*)lds);
  unsigned ar1 = lbase + (unsigned)(a1 * 6144 + th * 256 + r * 4 + 2 * hh) * 4u;
  unsigned ar2 = lbase + (unsigned)(a2 * 6144 + th * 256 + r * 4 + 2 * hh) * 4u;
  unsigned bro = lbase + (unsigned)(2 * W2_PX_FLOATS + j * 6 * 256 + hh * 64 + r * 2) * 4u;
  asm volatile("" : "+v"(ar1), "+v"(ar2), "+v"(bro));
  const float sg = j == 1 ? 1.f : -1.f;
  const f32x2 sgn = {sg, sg};
  const f32x2 c2 = {2.f, 2.f}, c4 = {4.f, 4.f}, c5 = {5.f, 5.f};
  constexpr unsigned WAIT_VM6_LGKM0 = 6 | 0x70, WAIT_VM0_LGKM0 = 0x70;
  // ---- this workgroup's items: whole blocks e = l, l + GL, ..., and — when the XCD's list leaves fewer than GL / 2 entries behind its last whole
  // round (layer1 at B = 32: 392 entries on 32 workgroups = 12 rounds + 8) — ONE COLUMN HALF of a left-over entry: two workgroups share such a
  // block, each runs its K loop for 32 of the 64 columns (12 instead of 24 MFMAs per stage and wave, the same pixel stages) and finishes its own
  // columns — no partial sums to exchange — so the launch ends half a block, not a whole one, behind its last whole round
  const int Rx = nent / GL, Lx = nent - Rx * GL;
  const bool halfmode = Lx > 0 && 2 * Lx <= GL;
  const int nwhole = Rx + ((!halfmode && l < Lx) ? 1 : 0);
  const bool has_half = halfmode && l < 2 * Lx;
  const int e_half = Rx * GL + (l >> 1), nh_half = l & 1;
  if (nwhole == 0 && !has_half) return;
  // first item: decode, first U stage and pixel double stage
  tflags = decode(nwhole ? l : e_half);
  nb_dma = entry_nb(nwhole ? l : e_half);
  issue_u(0, 0);
  issue_px(0, 0, 0);
  issue_px(0, 0, 1);
  W2P_STAMP_ID();
  [[maybe_unused]] int item = 0;
  // one item: HALF = only column half `nhsel` (accumulators, MFMAs, U reads and epilogue of that half); en >= 0 = the entry whose first stages
  // the epilogue prefetches
  auto run_item = [&](auto half_tag, const int e, const int nhsel, const int en) {
    constexpr bool HALF = decltype(half_tag)::value;
    constexpr int NHN = HALF ? 1 : 2;
    const int64_t tile0 = (int64_t)entry_mb(e) * 64;
    const int n0 = entry_nb(e) * 64;
    W2P_STAMP(item, 0);
    __builtin_amdgcn_s_setprio(0);
    f32x16 acc[6][NHN];
#pragma unroll
    for (int pp = 0; pp < 6; ++pp)
#pragma unroll
      for (int nh = 0; nh < NHN; ++nh)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[pp][nh][g] = 0.f;
    f32x2 V[6], ut[NHN];
#pragma unroll
    for (int pp = 0; pp < 6; ++pp) V[pp] = (f32x2){0.f, 0.f};
#pragma unroll
    for (int nh = 0; nh < NHN; ++nh) ut[nh] = (f32x2){0.f, 0.f};
    unsigned brh = bro + (unsigned)(HALF ? nhsel * 512 : 0);      // U reads: the column half's 512 bytes inside a point's block
    asm volatile("" : "+v"(brh));
    auto mfma_point = [&](const int pp, const f32x2 (&u)[NHN]) {
#pragma unroll
      for (int e2_ = 0; e2_ < 2; ++e2_)
#pragma unroll
        for (int nh = 0; nh < NHN; ++nh) acc[pp][nh] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[pp][e2_], u[nh][e2_], acc[pp][nh], 0, 0, 0);
    };
    auto read_u = [&](const int uslot, const int pp, f32x2 (&u)[NHN]) {
#pragma unroll
      for (int nh = 0; nh < NHN; ++nh) u[nh] = *(const __attribute__((address_space(3))) f32x2*)((lds_cptr)brh + (uslot * W2_U_FLOATS + pp * 256 + nh * 128) * 4);
    };
    for (int s0 = 0; s0 < NSL; s0 += 4) {
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) {
        const int sgl = s0 + sidx;
        const int e2 = sidx & 1, pslot = (sidx >> 1) & 1, uslot = sidx & 1;
        if (e2) __builtin_amdgcn_s_waitcnt(WAIT_VM6_LGKM0);
        else __builtin_amdgcn_s_waitcnt(WAIT_VM0_LGKM0);
        __builtin_amdgcn_s_barrier();
        const int pso = (pslot * 512 + e2 * 128) * 4;
        f32x2 d1[6], d2[6], cmb[6];
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          d1[b] = *(const __attribute__((address_space(3))) f32x2*)((lds_cptr)ar1 + pso + b * 4096);
          d2[b] = *(const __attribute__((address_space(3))) f32x2*)((lds_cptr)ar2 + pso + b * 4096);
        }
        f32x2 ua[NHN], ub[NHN];
        __builtin_amdgcn_sched_barrier(0);
        issue_u(sgl + 1, uslot ^ 1);
        mfma_point(5, ut);
        __builtin_amdgcn_sched_barrier(0);
        read_u(uslot, 0, ua);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 6; ++b) cmb[b] = pk_fma(d2[b], sgn, d1[b]);
        wino_bt6(cmb, V, c2, c4, c5);
        asm volatile("s_nop 1" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        read_u(uslot, 1, ub);
        mfma_point(0, ua);
        read_u(uslot, 2, ua);
        mfma_point(1, ub);
        __builtin_amdgcn_sched_barrier(0);
        if (!e2) issue_px((sgl >> 1) + 1, pslot ^ 1, 0);
        read_u(uslot, 3, ub);
        mfma_point(2, ua);
        read_u(uslot, 4, ua);
        mfma_point(3, ub);
        __builtin_amdgcn_sched_barrier(0);
        if (!e2) issue_px((sgl >> 1) + 1, pslot ^ 1, 1);
        read_u(uslot, 5, ut);
        mfma_point(4, ua);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    mfma_point(5, ut);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    W2P_STAMP(item, 1);
    // ---- epilogue.  The ring is dead; the piece offsets of the NEXT item go into the stash now (every DMA of this one has been issued)
    const bool more = en >= 0;                                   // workgroup-uniform
    int tflags_n = 0;
    if (more) tflags_n = decode(en);
    if (tid < 64) w2_tile_records(p, lds, tile0, tid);
    __builtin_amdgcn_s_setprio(SLIC_PRIO_EDGE);
#pragma unroll
    for (int nh = 0; nh < NHN; ++nh) {
      const f32x16 s12 = acc[1][nh] + acc[2][nh], d12 = acc[1][nh] - acc[2][nh];
      const f32x16 s34 = acc[3][nh] + acc[4][nh], d34 = acc[3][nh] - acc[4][nh];
      acc[0][nh] = acc[0][nh] + s12 + s34;
      acc[1][nh] = d12 + 2.f * d34;
      acc[2][nh] = s12 + 4.f * s34;
      acc[3][nh] = d12 + 8.f * d34 + acc[5][nh];
    }
    W2Stats st;
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      st.ref[nh] = st.sd[nh] = st.sq[nh] = st.s1[nh] = st.s2[nh] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    auto write_image = [&](const int nh) {
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int tl = th * 32 + (g & 3) + 8 * (g >> 2) + 4 * hh;
          lds[((j * 64 + tl) * 4 + o) * 32 + r] = acc[o][nh][g];
        }
    };
    const int n0a = n0 + (HALF ? nhsel * 32 : 0);                // first column of the item's first (or only) half
    // column half 0 (a half item: its only one)
    write_image(0);
    __syncthreads();
    if constexpr (LOADS) {
      w2p_passes_loads<0>(p, lds, n0a, wave, st);
    } else {
      W2Half h0;
      w2p_read(p, lds, n0a, wave, h0);
      w2p_write<0>(p, h0, st);
    }
    __syncthreads();                                             // the half's readers are done with the image
    if constexpr (!HALF) {
      // column half 1: without optional operands its rows are combined into registers first; behind the barrier that follows nothing reads the
      // image or the tile records any more, the ring may be written — the next item's first U stage and pixel double stage go out in FRONT of
      // this half's stores
      write_image(1);
      __syncthreads();
      [[maybe_unused]] W2Half h1;
      if constexpr (LOADS) w2p_passes_loads<1>(p, lds, n0 + 32, wave, st);
      else w2p_read(p, lds, n0 + 32, wave, h1);
      __syncthreads();
      W2P_STAMP(item, 2);
      if (more) {
        tflags = tflags_n;
        nb_dma = entry_nb(en);
        issue_u(0, 0);                                           // U slot 0 / pixel slot 0
        issue_px(0, 0, 0);
        issue_px(0, 0, 1);
      }
      W2P_STAMP(item, 3);
      if constexpr (!LOADS) w2p_write<1>(p, h1, st);
    }
    w2p_stats_finish<NHN>(p, lds, tile0 >> 6, n0a, wave, st);
    W2P_STAMP(item, 4);
    w2p_stats_merge<NHN>(p, tile0 >> 6, n0a, wave);
    W2P_STAMP(item, 5);
#ifdef SLIC_W2_STAMPS
    ++item;
#endif
  };
  for (int k = 0; k < nwhole; ++k) {
    const int e = l + k * GL;
    run_item(std::false_type{}, e, 0, k + 1 < nwhole ? e + GL : (has_half ? e_half : -1));
  }
  if (has_half) run_item(std::true_type{}, e_half, nh_half, -1);
}

// second pass of a K-split launch: one workgroup per (tail block, n block) adds the `pieces` pieces in order, lays the sums out as the
// epilogue expects its side-by-side contributions (row 0 as H-point 0, minus row 1 as H-point 3, H-points 1 and 2 zero) and runs it
__global__ __launch_bounds__(512) void conv_wino2_finish(const SlicConvArgs p, const int full_rows, const float* __restrict__ slab, const int mb_off,
                                                          const int pieces) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int mb = blockIdx.x, nb = blockIdx.y;
  const int64_t tile0 = (int64_t)(mb_off + mb) * 64;
  constexpr int JS = 64 * 4 * 32;                              // floats per H-point region (and per row of a slab piece)
  if (tid < 64) w2_tile_records(p, lds, tile0, tid);
  for (int nh = 0; nh < 2; ++nh) {
    if (nh) __syncthreads();
    const float* base = slab + (((int64_t)mb * gridDim.y + nb) * pieces * 2 + nh) * (2 * JS);
    for (int e = tid * 4; e < 2 * JS; e += 512 * 4) {          // e < JS: row 0; else row 1
      // every piece's load in flight at once (a loop of dependent adds paid a memory round trip per piece: with the 16 workgroups of a
      // small-batch layer4 launch that was most of the launch), then the adds in piece order
      f32x4 pv[16];
#pragma unroll
      for (int z = 0; z < 16; ++z)
        if (z < pieces) pv[z] = *(const f32x4*)(base + (int64_t)z * 2 * (2 * JS) + e);
      f32x4 v = pv[0];
#pragma unroll
      for (int z = 1; z < 16; ++z)
        if (z < pieces) v += pv[z];
      if (e < JS) {
        *(f32x4*)(lds + e) = v;
        *(f32x4*)(lds + JS + e) = (f32x4){0.f, 0.f, 0.f, 0.f};
      } else {
        *(f32x4*)(lds + 3 * JS + (e - JS)) = -v;
        *(f32x4*)(lds + 2 * JS + (e - JS)) = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    __syncthreads();
    w2_epilogue(p, lds, tile0, nb * 64 + nh * 32, tid, full_rows);
  }
}

// U2[(((((kt * C/4 + cc) * N/64 + nb) * 4 + j) * 6 + p) * 2 + nl / 32) * 2 + e / 2][nl % 32][e % 2]
//     = sum_kh sum_kw Gh[j][kh] Gw[p][kw] w(n = 64 nb + nl, c = 4 cc + e, kt, kh, kw)      (channel PAIR major inside a column half: see the kernel's reader offsets)
//   forward : w(n, c, kt, kh, kw) = W[n][c][kt][kh][kw]                    (N_ = out channels N, C_ = in channels C)
//   dgrad   : w(n, c, kt, kh, kw) = W[c][n][2 - kt][2 - kh][2 - kw]        (N_ = C: channels of dx, C_ = N: channels of dy)
// Gw (F(4, 3)) = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
// Gh (F(2, 3)) = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1], row 2 NEGATED (the kernel forms H-point 2 as d1 - d2)
__global__ void pack_w_wino2(const float* __restrict__ Wt, int N, int C, int dgrad, float* __restrict__ U) {
  const int N_ = dgrad ? C : N, C_ = dgrad ? N : C;
  const int64_t tot = (int64_t)3 * C_ * N_;
  const int64_t e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e0 >= tot) return;
  // e0 = (((kt * CCH + cc) * NB + nb) * 64 + nl) * 4 + e
  int64_t q = e0;
  const int e = (int)(q & 3); q >>= 2;
  const int nl = (int)(q & 63); q >>= 6;
  const int NB = N_ >> 6, CCH = C_ >> 2;
  const int nb = (int)(q % NB); q /= NB;
  const int cc = (int)(q % CCH); q /= CCH;
  const int kt = (int)q;
  const int n = nb * 64 + nl, c = cc * 4 + e;
  float w[3][3];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      if (dgrad) w[kh][kw] = Wt[((int64_t)c * C + n) * 27 + (2 - kt) * 9 + (2 - kh) * 3 + (2 - kw)];
      else w[kh][kw] = Wt[((int64_t)n * C + c) * 27 + kt * 9 + kh * 3 + kw];
    }
  // along H first: g[j][kw]
  float g[4][3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const float s02 = w[0][kw] + w[2][kw];
    g[0][kw] = w[0][kw];
    g[1][kw] = 0.5f * (s02 + w[1][kw]);
    g[2][kw] = -(0.5f * (s02 - w[1][kw]));
    g[3][kw] = w[2][kw];
  }
  const int64_t blk = ((int64_t)(kt * CCH + cc) * NB + nb) * (24 * 64 * 4);
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const float s02 = g[jj][0] + g[jj][2];
    float u[6];
    u[0] = 0.25f * g[jj][0];
    u[1] = (-1.f / 6.f) * (s02 + g[jj][1]);
    u[2] = (-1.f / 6.f) * (s02 - g[jj][1]);
    u[3] = (1.f / 24.f) * g[jj][0] + (1.f / 12.f) * g[jj][1] + (1.f / 6.f) * g[jj][2];
    u[4] = (1.f / 24.f) * g[jj][0] - (1.f / 12.f) * g[jj][1] + (1.f / 6.f) * g[jj][2];
    u[5] = g[jj][2];
#pragma unroll
    for (int pp = 0; pp < 6; ++pp) U[blk + ((((jj * 6 + pp) * 2 + (nl >> 5)) * 2 + (e >> 1)) * 32 + (nl & 31)) * 2 + (e & 1)] = u[pp];
  }
}

// real outputs in a full block of 64 tiles, or 0 when the blocks of this geometry do not all hold the same number
int slic_wino2_full_rows(const SlicConvArgs* a) {
  const int W = a->Ws, H = a->Hs;
  const int Wq = (W + 3) / 4, Hq = (H + 1) / 2;
  if (W % 4 == 0 && H % 2 == 0) return 512;
  if (H % 2 == 0) return 64 % Wq == 0 ? (64 / Wq) * 2 * W : 0;
  return 64 % (Hq * Wq) == 0 ? (64 / (Hq * Wq)) * H * W : 0;
}

static int wino2_check(const SlicConvArgs* a, int* full) {
  SLIC_REQUIRE(a->Cs % 64 == 0 && a->N % 64 == 0 && a->sa == 1 && a->sb == 1 && a->sc == 1 && a->Ga == a->Ts && a->Gb == a->Hs &&
                   a->Gc == a->Ws && !a->dst_strided && !a->bias && !a->k_run_len,
               "slic_conv_gemm: variant 31 needs a stride-1 same-size geometry, Cs %% 64 == 0, N %% 64 == 0, no bias");
  SLIC_REQUIRE(((a->Cs / 4) & (a->Cs / 4 - 1)) == 0, "slic_conv_gemm: variant 31 needs Cs / 4 to be a power of two");
  SLIC_REQUIRE((uint64_t)a->wgt_bytes == (uint64_t)3 * 24 * a->Cs * a->N * 4, "slic_conv_gemm: variant 31: wgt_bytes != 3 * 24 * Cs * N floats");
  *full = slic_wino2_full_rows(a);
  SLIC_REQUIRE(*full > 0, "slic_conv_gemm: variant 31: blocks of 64 tiles hold different numbers of outputs at H=%d W=%d", a->Hs, a->Ws);
  const int64_t tiles = (a->M / ((int64_t)a->Hs * a->Ws)) * ((a->Hs + 1) / 2) * ((a->Ws + 3) / 4);
  // the kernel's resource spans the tensor plus a frame on either side (kt rides in the scalar offset, which the range check does not see): its
  // size must stay below the out-of-range vector offset 0xFFFFFF00 that stands for a padding pixel, and must not wrap in 32 bits
  SLIC_REQUIRE(tiles < (1ll << 31) && (int64_t)a->M * a->Cs * 4 + 2 * (int64_t)a->Hs * a->Ws * a->Cs * 4 + 16 <= 0xFFFFFF00ll,
               "slic_conv_gemm: variant 31: tensor too large (source + two frames must stay below 4 GiB - 256 B)");
  return SLIC_OK;
}

static constexpr size_t w2_lds_bytes() {
  constexpr size_t ring = (size_t)W2_RING_FLOATS * sizeof(float) + 512 * 8 * 4, epi = (size_t)conv_epi_lds_floats(512, 64, 8) * sizeof(float);
  return ring > epi ? ring : epi;
}

// a K-split piece is a whole number of loop turns (two double stages): pieces | 3 Cs / 16
static bool wino2_pieces_ok(const SlicConvArgs* a, int pieces) { return pieces >= 2 && pieces <= 16 && (3 * a->Cs / 16) % pieces == 0; }

size_t slic_conv_wino2_split_workspace_bytes(const SlicConvArgs* a, int nfull, int pieces) {
  const int64_t tiles = (a->M / ((int64_t)a->Hs * a->Ws)) * ((a->Hs + 1) / 2) * ((a->Ws + 3) / 4);
  const int64_t tail = slic_cdiv(tiles, 64) - nfull;
  return tail <= 0 || !wino2_pieces_ok(a, pieces) ? 0 : slic_align_up((size_t)tail * (a->N / 64) * pieces * 2 * (2 * 64 * 4 * 32) * sizeof(float), 256);
}

// compute units of the current device (one 160 KB workgroup each: the persistent kernel's grid); SLIC_WINO2_PERSIST_GRID overrides (tests:
// a grid of 8 makes small shapes walk lists of several blocks)
static int w2_persist_grid() {
  const char* e = getenv("SLIC_WINO2_PERSIST_GRID");
  if (e && atoi(e) >= 8) return atoi(e) / 8 * 8;
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
    n = pr.multiProcessorCount >= 8 ? pr.multiProcessorCount / 8 * 8 : 8;
  }
  return n;
}

// SLIC_WINO2_PERSIST: 0 = the one-block-per-workgroup kernel everywhere; 1 (default) = the whole tile blocks of a launch with at least two
// blocks per compute unit go to the persistent kernel (conv_wino2p_kernel); 2 = only the launches that emit forward BatchNorm statistics
// (a training step's forward: nothing shares the chip with them; the data gradients run beside the side stream's weight gradients)
static int w2_persist_mode() {
  const char* e = getenv("SLIC_WINO2_PERSIST");
  return e ? atoi(e) : 1;
}

// how many of the tile blocks [0, nblocks) go to the persistent kernel: the whole ones (all of them when the tile count is a multiple of
// 64), or 0 when the launch is not eligible.  full = real outputs of a full block (512 = whole 2 x 4 tiles everywhere)
static int w2_persist_blocks(const SlicConvArgs* a, int full, int64_t tiles, int nblocks, unsigned ny) {
  const int mode = w2_persist_mode();
  if (mode <= 0 || (mode == 2 && !a->stat_partial)) return 0;
  if (full != 512 || (a->Ws & 3) || (a->Hs & 1)) return 0;
  const int64_t whole = tiles / 64 < nblocks ? tiles / 64 : nblocks;
  return whole * ny >= 2ll * w2_persist_grid() ? (int)whole : 0;
}

// dynamic LDS of the persistent kernel: the rings + the piece offsets (+ the epilogue image, inside the rings); 4 KB of static reduction rows beside it
constexpr size_t W2P_LDS_BYTES = (size_t)W2_RING_FLOATS * sizeof(float) + 6 * 512 * 4;
static_assert(W2P_LDS_BYTES + 4096 <= 160 * 1024 && W2P_LDS_BYTES >= (size_t)(W2_TREC_FLOATS + 128) * sizeof(float), "LDS");

static int w2_launch_persist(const SlicConvArgs* a, hipStream_t st, int nblocks, unsigned ny, size_t lds) {
  (void)lds;
  const bool loads = a->addend || a->mask_src || a->bwd_z || a->scale || a->shift;
  const dim3 grid((unsigned)w2_persist_grid());
#define W2P_GO(L, C_) conv_wino2p_kernel<L, C_><<<grid, dim3(512), W2P_LDS_BYTES, st>>>(*a, nblocks, (int)ny)
#define W2P_PICK(L)                                                                    \
  do {                                                                                  \
    if (a->Cs == 64) W2P_GO(L, 64); else if (a->Cs == 128) W2P_GO(L, 128);               \
    else if (a->Cs == 256) W2P_GO(L, 256); else W2P_GO(L, 0);                            \
  } while (0)
  if (loads) W2P_PICK(true); else W2P_PICK(false);
#undef W2P_PICK
#undef W2P_GO
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

// nfull < 0: the plain launch.  nfull >= 0: tile blocks [0, nfull) whole, the rest with the K loop cut into `pieces` + the finish pass.
int slic_conv_wino2_launch(const SlicConvArgs* a, hipStream_t st, int nfull, float* slab, int pieces) {
  int full;
  int rc = wino2_check(a, &full);
  if (rc) return rc;
  constexpr size_t lds = w2_lds_bytes();
  static_assert(lds <= 160 * 1024, "LDS");
  static bool attr_set = false;
  if (!attr_set) {
    SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)conv_wino2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#define W2P_ATTR(L, C_) SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)conv_wino2p_kernel<L, C_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W2P_LDS_BYTES))
    W2P_ATTR(false, 0); W2P_ATTR(false, 64); W2P_ATTR(false, 128); W2P_ATTR(false, 256);
    W2P_ATTR(true, 0); W2P_ATTR(true, 64); W2P_ATTR(true, 128); W2P_ATTR(true, 256);
#undef W2P_ATTR
    SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)conv_wino2_finish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int64_t tiles = (a->M / ((int64_t)a->Hs * a->Ws)) * ((a->Hs + 1) / 2) * ((a->Ws + 3) / 4);
  const int gx = (int)slic_cdiv(tiles, 64);
  const unsigned ny = (unsigned)(a->N / 64);
  if (nfull < 0 || nfull >= gx) {
    const int np = w2_persist_blocks(a, full, tiles, gx, ny);
    if (np > 0) {
      rc = w2_launch_persist(a, st, np, ny, lds);
      if (rc || np == gx) return rc;
    }
    // (np < gx: the last, partly filled block — a tile count that is not a multiple of 64 — on the one-block kernel)
    conv_wino2_kernel<<<dim3((unsigned)((gx - np + 7) / 8 * 8), ny), dim3(512), lds, st>>>(*a, full, nullptr, np, gx - np);
    SLIC_LAUNCH_CHECK();
    return SLIC_OK;
  }
  SLIC_REQUIRE(slab, "slic_conv_gemm_tailsplit: variant 31 needs a workspace");
  SLIC_REQUIRE(wino2_pieces_ok(a, pieces), "slic_conv_gemm_tailsplit: variant 31: splits = %d must divide 3 Cs / 16 = %d (2 .. 16)", pieces, 3 * a->Cs / 16);
  if (nfull > 0) {
    const int np = w2_persist_blocks(a, full, tiles, nfull, ny);
    if (np > 0) {
      rc = w2_launch_persist(a, st, np, ny, lds);
      if (rc) return rc;
    }
    if (np < nfull) {
      conv_wino2_kernel<<<dim3((unsigned)((nfull - np + 7) / 8 * 8), ny), dim3(512), lds, st>>>(*a, full, nullptr, np, nfull - np);
      SLIC_LAUNCH_CHECK();
    }
  }
  const int tail = gx - nfull;
  conv_wino2_kernel<<<dim3((unsigned)((tail + 7) / 8 * 8), ny, (unsigned)pieces), dim3(512), lds, st>>>(*a, full, slab, nfull, tail);
  SLIC_LAUNCH_CHECK();
  conv_wino2_finish<<<dim3((unsigned)tail, ny), dim3(512), lds, st>>>(*a, full, slab, nfull, pieces);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_pack_weight_wino2(const float* W, int N, int C, int dgrad, float* U, void* stream) {
  SLIC_REQUIRE(W && U && N % 64 == 0 && C % 64 == 0, "slic_pack_weight_wino2: N and C must be multiples of 64");
  const int64_t tot = (int64_t)3 * C * N;
  pack_w_wino2<<<dim3((unsigned)slic_cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream>>>(W, N, C, dgrad, U);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

// ------------------------------------------------------------------------------------------
// Weight gradient of the same layers by the TRANSPOSED two-dimensional algorithm: with Y = Ah^T [(Gh w Gw^T) . (Bh^T x Bw)] Aw per tile,
//   dL/dw = Gh^T [ sum over tiles of (Bh^T x Bw) . (Ah dY Aw^T) ] Gw
// — the forward's input transform V (24 points from the tile's 4 x 6 patch), the output transform run backwards Z (24 points from the
// tile's 2 x 4 output gradients), one 64 x 64 product-sum per point, kt and tile, and Gh^T .. Gw once at the very end: 24 multiplies per
// (kt, c, n) and tile of 8 outputs where the one-dimensional transposed algorithm has 36 and the direct form 72.
//   Workgroup = 512 threads = one kt, one 64 c x 64 n block, one slice of the tiles, ALL FOUR H-points: wave (j, wc) owns H-point j of
//   32 channels for all 64 columns — 6 W-points x 2 column halves = 12 accumulators, the forward kernel's budget.  The MFMA's k dimension
//   is the TILE: lane (r, hh) reads its channel's pixels of tile 2 ks + hh (two patch rows x 6) and its column's gradients in both column
//   halves (the even and the odd columns of the block: two output rows x 4, one ds_read_b64 each), forms the H-point, runs the F(4, 3) transforms along W in registers — V in scalars
//   (one k-step), Z packed over the two column halves — and feeds two MFMAs per W-point.
//   Stage = 4 tiles (two k-steps), tile image [24 pixels][64 ch] + [8 gradients][64 n] = 8 KB (whole cache lines: 170 bytes per MFMA),
//   4-stage ring (128 KB: one workgroup per CU), wave w DMAs four of the eight 1 KB pieces of tile w / 2; counted vmcnt, one barrier per
//   stage; k-step 1 of a stage is multiplied behind the NEXT stage's barrier, under the latency of its first reads.  Per-tile records
//   {pixel index, invalid bits} from slic_conv_wino2_tile_table come by SCALAR loads, a stage ahead (a vector load would sit in the
//   in-order vmcnt queue behind the DMAs: waiting for the record then drains the ring — the round's first kernel, one H-point PAIR per
//   workgroup with 32 c x 32 n waves, did exactly that and ran at 0.50 of the pipe; this form: 0.64 at layer1).
//   Slabs [slice][kt][j][kw][c][n] (Gw^T applied by the wave that holds the six W-points); conv_wgrad_wino2_sum adds groups of slices in
//   order, conv_wgrad_wino2_reduce adds the groups and applies Gh^T.
// ------------------------------------------------------------------------------------------
// tab[tile] = {pixel index of (b, t, 2 h2, 4 wt); bits 0-23: patch pixel (a, b) = (2 h2 - 1 + a, 4 wt - 1 + b) is OUTSIDE the frame
//              (bit a * 6 + b); bits 24-26: frame t - 1 + kt is outside the clip; bit 27: always set (a record read past the slice is all zeros)}
__global__ void conv_wino2_tile_table_kernel(const SlicConvArgs p, uint2* __restrict__ tab) {
  const int Wq = (p.Ws + 3) >> 2, Hq = (p.Hs + 1) >> 1;
  const int64_t tile = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tile >= (p.M / ((int64_t)p.Hs * p.Ws)) * Hq * Wq) return;
  unsigned q = (unsigned)tile;
  const int wt = (int)(q % (unsigned)Wq); q /= (unsigned)Wq;
  const int h2 = (int)(q % (unsigned)Hq); q /= (unsigned)Hq;
  const int tt = (int)(q % (unsigned)p.Ts);
  unsigned mk = 1u << 27;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const bool in = (unsigned)(2 * h2 - 1 + a) < (unsigned)p.Hs && (unsigned)(4 * wt - 1 + b) < (unsigned)p.Ws;
      mk |= (in ? 0u : 1u) << (a * 6 + b);
    }
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) mk |= ((unsigned)(tt + kt - 1) < (unsigned)p.Ts ? 0u : 1u) << (24 + kt);
  tab[tile] = make_uint2((unsigned)(((int64_t)q * p.Hs + 2 * h2) * p.Ws + 4 * wt), mk);
}

constexpr int WB_TILE_BYTES = 8192;
constexpr int WB_STAGE_BYTES = 4 * WB_TILE_BYTES;
#ifndef SLIC_WB_STAGES
#define SLIC_WB_STAGES 4
#endif
constexpr int WB_STAGES = SLIC_WB_STAGES;

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_wgrad_wino2_kernel(const SlicConvArgs p, const float* __restrict__ dy, unsigned dy_bytes, const uint2* __restrict__ tile_tab,
                              float* __restrict__ slab, int tiles_per_split, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = wave & 3, wc = wave >> 2;
  const int r = lane & 31, hh = lane >> 5;
  const int C = p.Cs, N = p.N, H = p.Hs, W = p.Ws;
  const int CB = C >> 6, NBk = N >> 6;
  const int per_slice = 3 * CB * NBk;
  const int bx = blockIdx.x, gdx = gridDim.x;
  const int v = (bx & 7) * (gdx >> 3) + (bx >> 3);            // XCD-aware: the workgroups of a slice (same tiles) share an L2
  if (v >= per_slice * nsplit) return;
  const int z = v / per_slice;
  int rest = v - z * per_slice;
  const int kt = rest / (CB * NBk); rest -= kt * (CB * NBk);
  const int cb = rest / NBk, nb = rest - cb * NBk;
  const int Wq = (W + 3) >> 2, Hq = (H + 1) >> 1;
  const int64_t Mt = (p.M / ((int64_t)H * W)) * Hq * Wq;
  const int64_t tbeg = (int64_t)z * tiles_per_split;
  const int64_t tend = min(tbeg + tiles_per_split, Mt);
  const int nst = tend > tbeg ? (int)((tend - tbeg + 3) / 4) : 0;
  __builtin_amdgcn_s_setprio(SLIC_PRIO_EDGE);
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);
  // ---- DMA roles: wave w serves tile dt = w / 2 of the stage with pieces d = 4 (w % 2) + i, i = 0..3 (d 0-5: patch pixels 4 d .. 4 d + 3,
  // pixel e = a * 6 + b; d 6, 7: the gradients of output row d - 6); lane = (slot within the piece lane / 16, 16-byte chunk lane % 16 of the
  // slot's 64 channels / columns); piece d lands at tile byte 1024 d.  Pieces 0, 1 of a wave are always pixels; pieces 2, 3 are pixels
  // for even waves and gradients for odd ones: their resource, row pitch and bias are wave-uniform selections made once.
  const int dt = wave >> 1, dh = wave & 1;
  const int sub = lane >> 4, chunk = lane & 15;
  // Fast path (every pixel of the patch and frame t - 1 + kt exist): the tile's base offset is wave-uniform and rides in the DMA's SCALAR
  // offset.  The per-lane constants are biased to be non-negative (B0x: one frame for kt = 0, one row, one pixel), the scalar part
  // carries the rest (never negative when the pixels exist).
  const unsigned B0x = (unsigned)((((kt == 0 ? H * W : 0) + W + 1) * C) * 4);
  const __amdgpu_buffer_rsrc_t rs_23 = __builtin_amdgcn_make_buffer_rsrc(dh ? (void*)dy : (void*)p.src, 0, dh ? (int)dy_bytes : (int)p.src_bytes, 0x00020000);
  const unsigned pitch23 = (unsigned)((dh ? N : C) * 4), bias23 = dh ? 0u : B0x;
  unsigned cstb[4];                                            // per-lane constant part of the piece's byte offset, + its bias
  int spos[4];                                                 // validity bit of the slot = spos + sub (wave-uniform part)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int d = 4 * dh + i;
    if (d >= 6) {
      const int hp = d - 6, o = sub;
      spos[i] = (hp + 1) * 6 + 1;                              // output (hp, o) is patch pixel (hp + 1, o + 1)
      cstb[i] = (unsigned)(((hp * W + o) * N + nb * 64 + chunk * 4) * 4);
    } else {
      const int e = 4 * d + sub, a = e / 6, b = e - 6 * a;
      spos[i] = 4 * d;
      cstb[i] = (unsigned)(((((kt - 1) * H + (a - 1)) * W + (b - 1)) * C + cb * 64 + chunk * 4) * 4) + B0x;
    }
  }
  const unsigned needx = 0x00FFFFFFu | (1u << (24 + kt));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  // the record of this wave's tile of stage s — a SCALAR load (the tile is wave-uniform): it does not enter the vector-memory queue, whose
  // in-order counter would otherwise make the wait for a record a wait for every DMA issued before it
  auto load_rec = [&](int s) -> u32x2 {
    const int64_t tile = tbeg + (int64_t)s * 4 + dt;
    u32x2 rc = {0u, 0u};                                       // past the slice: zeros -> dead below
    if (tile < tend) { const uint2 t = tile_tab[tile]; rc.x = t.x; rc.y = t.y; }
    return rc;
  };
  auto piece_dst = [&](int slot_, int i) {
    return (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) char*)lds + slot_ * WB_STAGE_BYTES + dt * WB_TILE_BYTES +
                                                      (4 * dh + i) * 1024);
  };
  // pieces i0, i0 + 1 (i0 = 0 or 2) of the tile whose record is rc (wave-uniform)
  auto issue_tile = [&](int slot_, u32x2 rc, const int i0) {
    const unsigned ry = (unsigned)__builtin_amdgcn_readfirstlane((int)rc.y), rx = (unsigned)__builtin_amdgcn_readfirstlane((int)rc.x);
    const unsigned pitch = i0 ? pitch23 : (unsigned)(C * 4), bias = i0 ? bias23 : B0x;
    const bool isy = i0 && dh;
    if ((ry & needx) == 0 && ((ry >> 27) & 1u)) {
      const unsigned so_ = rx * pitch - bias;
#pragma unroll
      for (int i = i0; i < i0 + 2; ++i) {
#if SLIC_W2_ABL & 1
        __builtin_amdgcn_raw_ptr_buffer_load_lds(i0 ? rs_23 : rs_src, piece_dst(slot_, i), 16, (int)(0xFFFFFF00u + 0 * cstb[i]), 0, 0, 0);
#else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(i0 ? rs_23 : rs_src, piece_dst(slot_, i), 16, (int)cstb[i], (int)so_, 0, 0);
#endif
      }
    } else {
      // invalid slot -> offset | -1 (out of range: zeros).  A record past the slice reads {0, 0}: bit 27 is clear there
      const bool dead = !((ry >> 27) & 1u);
      const unsigned tinv = (unsigned)__builtin_amdgcn_sbfe((int)ry, 24 + kt, 1);          // -1: frame t - 1 + kt outside
      const unsigned m = dead ? 0xFFFFFFFFu : (isy ? ry : (ry | tinv));
      const unsigned base = rx * pitch - bias;
#pragma unroll
      for (int i = i0; i < i0 + 2; ++i) {
        const unsigned inv = (unsigned)(-(int)((m >> (unsigned)(spos[i] + sub)) & 1u));
        const unsigned off = (base + cstb[i]) | inv;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(i0 ? rs_23 : rs_src, piece_dst(slot_, i), 16, (int)off, 0, 0, 0);
      }
    }
  };
  f32x16 acc[6][2];
#pragma unroll
  for (int pp = 0; pp < 6; ++pp)
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[pp][nh][g] = 0.f;
  u32x2 recn;
#pragma unroll
  for (int t = 0; t < WB_STAGES - 1; ++t) {
    const u32x2 rc = load_rec(t);
    issue_tile(t, rc, 0);
    issue_tile(t, rc, 2);
    asm volatile("" ::: "memory");
  }
  recn = load_rec(WB_STAGES - 1);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_setprio(0);
  // ---- readers.  H-point j: V = x[a1] + sx * x[a2] (patch rows),  Z = y[ya] + sy * y[1]:
  //   j = 0: rows (0, 2), sx = -1; Z = y0            j = 1: rows (1, 2), sx = +1; Z = y0 + y1
  //   j = 2: rows (1, 2), sx = -1 (the point NEGATED); Z = y0 - y1      j = 3: rows (1, 3), sx = -1; Z = y1 (NEGATED)
  // (the two negations are undone by the reduce kernel's Gh)
  const int a1 = j == 0 ? 0 : 1, a2 = j == 3 ? 3 : 2;
  const float sxf = j == 1 ? 1.f : -1.f, syf = j == 1 ? 1.f : -1.f;
  const f32x2 sy = {syf, syf};
  const int ya = j == 3 ? 1 : 0;
  const bool two_rows = j == 1 || j == 2;
  // lane addresses (bytes; + slot * WB_STAGE_BYTES + ks * 2 * WB_TILE_BYTES): tile 2 ks + hh, this lane's channel / column
  typedef __attribute__((address_space(3))) const char* lds_cptr;
  const unsigned lbase = (unsigned)(size_t)((__attribute__((address_space(3))) float*)lds);
  unsigned xr1 = lbase + (unsigned)(hh * WB_TILE_BYTES + a1 * 1536 + (wc * 32 + r) * 4);      // + 256 b
  unsigned xr2 = lbase + (unsigned)(hh * WB_TILE_BYTES + a2 * 1536 + (wc * 32 + r) * 4);
  // the wave's two column "halves" are the EVEN and the ODD columns of the block (lane r: columns 2 r, 2 r + 1): one ds_read_b64 per
  // gradient instead of two dword reads 128 bytes apart — the kernel's LDS pipe was ~80 % busy with dword reads
  unsigned yr1 = lbase + (unsigned)(hh * WB_TILE_BYTES + 6144 + ya * 1024 + r * 8);             // + 256 o
  unsigned yr2 = lbase + (unsigned)(hh * WB_TILE_BYTES + 6144 + 1024 + r * 8);
  asm volatile("" : "+v"(xr1), "+v"(xr2), "+v"(yr1), "+v"(yr2));
  const f32x2 c4 = {4.f, 4.f}, c8 = {8.f, 8.f};
  // operands of one k-step: V (six W-points of this lane's channel) and Z (six W-points of its column in both column halves)
  struct Ops { float V[6]; f32x2 Z[6]; };
  auto read_x = [&](const int so, float (&xa)[6], float (&xb)[6]) {
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      xa[b] = *(const __attribute__((address_space(3))) float*)((lds_cptr)xr1 + so + b * 256);
      xb[b] = *(const __attribute__((address_space(3))) float*)((lds_cptr)xr2 + so + b * 256);
    }
  };
  auto read_y = [&](const int so, f32x2 (&yf)[4], f32x2 (&ys)[4]) {
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      yf[o] = *(const __attribute__((address_space(3))) f32x2*)((lds_cptr)yr1 + so + o * 256);
      ys[o] = *(const __attribute__((address_space(3))) f32x2*)((lds_cptr)yr2 + so + o * 256);
    }
  };
  // H-point, then V = B^T (.) along W (scalars: one k-step of this lane's channel)
  auto transform_x = [&](const float (&xa)[6], const float (&xb)[6], Ops& t) {
    __builtin_amdgcn_sched_barrier(0);
    float d[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) d[b] = __builtin_fmaf(sxf, xb[b], xa[b]);
    const float t1 = __builtin_fmaf(-4.f, d[2], d[4]), t2 = __builtin_fmaf(-4.f, d[1], d[3]);
    const float t3 = d[4] - d[2], u = d[3] - d[1];
    t.V[0] = __builtin_fmaf(4.f, d[0], __builtin_fmaf(-5.f, d[2], d[4]));
    t.V[1] = t1 + t2;
    t.V[2] = t1 - t2;
    t.V[3] = __builtin_fmaf(2.f, u, t3);
    t.V[4] = __builtin_fmaf(-2.f, u, t3);
    t.V[5] = __builtin_fmaf(4.f, d[1], __builtin_fmaf(-5.f, d[3], d[5]));
    __builtin_amdgcn_sched_barrier(0);
  };
  // H-point, then Z = A (.) along W for both column halves at once (packed): Z = [c0, c0+c1+c2+c3, c0-c1+c2-c3, c0+2c1+4c2+8c3,
  // c0-2c1+4c2-8c3, c3]; one fenced block closed by the two wait states an MFMA needs behind the (inline-assembly) op that wrote its operand
  auto transform_y = [&](const f32x2 (&yf)[4], const f32x2 (&ys)[4], Ops& t) {
    __builtin_amdgcn_sched_barrier(0);
    f32x2 cz[4];
    if (two_rows) {                                            // wave-uniform: H-points 1, 2 combine both gradient rows
#pragma unroll
      for (int o = 0; o < 4; ++o) cz[o] = pk_fma(ys[o], sy, yf[o]);
    } else {
#pragma unroll
      for (int o = 0; o < 4; ++o) cz[o] = yf[o];
    }
    const f32x2 e = pk_add(cz[0], cz[2]), od = pk_add(cz[1], cz[3]);
    const f32x2 e4 = pk_fma(cz[2], c4, cz[0]), o4 = pk_fma(cz[3], c8, pk_add(cz[1], cz[1]));
    t.Z[0] = cz[0];
    t.Z[1] = pk_add(e, od);
    t.Z[2] = pk_sub(e, od);
    t.Z[3] = pk_add(e4, o4);
    t.Z[4] = pk_sub(e4, o4);
    t.Z[5] = cz[3];
    asm volatile("s_nop 1" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // the MFMAs of W-points [p0, p1) of a transformed k-step (two column halves each)
  auto mfma_pts = [&](const Ops& t, const int p0, const int p1) {
#pragma unroll
    for (int pp = p0; pp < p1; ++pp)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) acc[pp][nh] = __builtin_amdgcn_mfma_f32_32x32x2f32(t.V[pp], t.Z[pp][nh], acc[pp][nh], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  Ops prev;
#pragma unroll
  for (int pp = 0; pp < 6; ++pp) { prev.V[pp] = 0.f; prev.Z[pp] = (f32x2){0.f, 0.f}; }
  constexpr int PER = 4;                                       // vector-memory ops per stage and wave: four DMAs
  constexpr int VMW = (WB_STAGES - 2) * PER;                   // 8
  auto stage = [&](const int sg, auto sl_) {
    constexpr int sidx = decltype(sl_)::value;
    // stage sg has landed once only the DMAs of the STAGES - 2 stages after it are outstanding; and this wave's own LDS reads of the
    // previous stage are complete
    __builtin_amdgcn_s_waitcnt((VMW & 0xF) | 0x0070 | (((VMW >> 4) & 3) << 14));             // vmcnt(8) lgkmcnt(0) expcnt(7)
#if !(SLIC_W2_ABL & 2)
    __builtin_amdgcn_s_barrier();
#endif
    constexpr int slotn = (sidx + WB_STAGES - 1) % WB_STAGES;
    constexpr int so = sidx * WB_STAGE_BYTES;
    float xa[6], xb[6];
    f32x2 yf[4], ys[4];
    Ops cur;
    // k-step 0 of this stage: its reads and transforms between the MFMAs of the previous stage's k-step 1
    read_x(so, xa, xb);
    __builtin_amdgcn_sched_barrier(0);
    mfma_pts(prev, 0, 3);
    read_y(so, yf, ys);
    transform_x(xa, xb, cur);
    mfma_pts(prev, 3, 6);
    transform_y(yf, ys, cur);
    // k-step 1: the same between the MFMAs of k-step 0; the stage's DMAs (the slot freed by the barrier above) among them
    read_x(so + 2 * WB_TILE_BYTES, xa, xb);
    __builtin_amdgcn_sched_barrier(0);
    mfma_pts(cur, 0, 3);
    issue_tile(slotn, recn, 0);
    read_y(so + 2 * WB_TILE_BYTES, yf, ys);
    transform_x(xa, xb, prev);
    mfma_pts(cur, 3, 6);
    issue_tile(slotn, recn, 2);
    recn = load_rec(sg + WB_STAGES);
    transform_y(yf, ys, prev);
  };
  for (int s0 = 0; s0 < nst; s0 += WB_STAGES) {
    stage(s0, std::integral_constant<int, 0>{});
    stage(s0 + 1, std::integral_constant<int, 1>{});
    stage(s0 + 2, std::integral_constant<int, 2>{});
    stage(s0 + 3, std::integral_constant<int, 3>{});
    if constexpr (WB_STAGES == 5) stage(s0 + 4, std::integral_constant<int, 4>{});
  }
  mfma_pts(prev, 0, 6);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_setprio(SLIC_PRIO_EDGE);
  // slab[z][kt][j][kw][c][n]: the six W-points of an H-point sit in this wave's accumulators, so they go back through Gw^T here, before the
  // slices are summed (both are linear) — the slabs, the summing pass and the reduce pass move 36 values per (c, n) instead of 72
  // Gw^T rows: kw 0: [1/4 -1/6 -1/6 1/24 1/24 0], kw 1: [0 -1/6 1/6 1/12 -1/12 0], kw 2: [0 -1/6 -1/6 1/6 1/6 1]
  float* out = slab + (((int64_t)z * 3 + kt) * 4 + j) * 3 * (int64_t)C * N + nb * 64 + 2 * r;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int c = cb * 64 + 32 * wc + (g & 3) + 8 * (g >> 2) + 4 * hh;
    f32x2 t[3];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      const float s12 = acc[1][nh][g] + acc[2][nh][g], d12 = acc[2][nh][g] - acc[1][nh][g];
      const float s34 = acc[3][nh][g] + acc[4][nh][g], d34 = acc[3][nh][g] - acc[4][nh][g];
      t[0][nh] = 0.25f * acc[0][nh][g] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
      t[1][nh] = (1.f / 6.f) * d12 + (1.f / 12.f) * d34;
      t[2][nh] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + acc[5][nh][g];
    }
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) *(f32x2*)(out + ((int64_t)kw * C + c) * N) = t[kw];      // columns 2 r, 2 r + 1: 256-byte runs per half wave
  }
}

// dW[n][c][kt][kh][kw] = sum_j Gh[j][kh] * (sum over the G partial slabs g, ascending, of slab[g * gstride][kt][j][kw][c][n])
// Gh as the kernel left the points: rows 2 and 3 negated — [1 0 0; 1/2 1/2 1/2; -1/2 1/2 -1/2; 0 0 -1]
// Workgroup = 8 channels x 32 columns, all three kt: a thread reads the 36 sums of its (c, n) — columns fastest: 128-byte runs —
// and the 27 weights leave through LDS, so that the stores are 864-byte runs of dW's [n][c 8][27] (a thread writing its own 9 values
// wrote 4 bytes every 27 C floats: 210 us for layer4's 75 MB).
// The loads of one kt — 12 values x G partial slabs — are written out unrolled so that they are all in flight together: a launch of
// 16 workgroups (layer1) is latency and nothing else, and the earlier form (a run-time slice loop inside the point loops: one
// round trip per point) took 75 us for 1.2 MB.  G <= 8 partial slabs: the slices themselves where the tile dimension is cut few ways (layers 3 / 4: no
// separate summing pass), the groups conv_wgrad_wino2_sum leaves otherwise.
template <int G>
__global__ __launch_bounds__(256) void conv_wgrad_wino2_reduce(const float* __restrict__ slab, int64_t gstride, int C, int N, float* __restrict__ dW) {
  __shared__ float outs[32 * 217];                               // [n 32][c 8][27], rows padded to 217 floats (bank spread)
  const int tid = threadIdx.x;
  const int nl = tid & 31, cl = tid >> 5;
  const int c0 = blockIdx.x * 8, n0 = blockIdx.y * 32;
  const int n = n0 + nl, c = c0 + cl;
  const int64_t CN = (int64_t)C * N;
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) {
    float v[4][3][G];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float* q = slab + (((int64_t)kt * 4 + jj) * 3 + kw) * CN + (int64_t)c * N + n;
#pragma unroll
        for (int g = 0; g < G; ++g) v[jj][kw][g] = q[g * gstride];
      }
    float t[4][3];      // [j][kw]
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        float a = v[jj][kw][0];
#pragma unroll
        for (int g = 1; g < G; ++g) a += v[jj][kw][g];           // partial slabs in order
        t[jj][kw] = a;
      }
    float* o = outs + nl * 217 + cl * 27 + kt * 9;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const float h12 = 0.5f * (t[1][kw] - t[2][kw]);            // 1/2 (S1 + S2) with S2 stored negated
      o[0 * 3 + kw] = t[0][kw] + h12;
      o[1 * 3 + kw] = 0.5f * (t[1][kw] + t[2][kw]);              // 1/2 (S1 - S2)
      o[2 * 3 + kw] = h12 - t[3][kw];                            // + S3, stored negated
    }
  }
  __syncthreads();
  for (int i = tid; i < 32 * 216; i += 256) {
    const int row = i / 216, e = i - row * 216;
    dW[((int64_t)(n0 + row) * C + c0) * 27 + e] = outs[row * 217 + e];
  }
}

// Group g = blockIdx.y of the slices: slab[g * per][e] = sum over slices z = g * per .. min(S, (g + 1) * per) - 1, ascending, of slab[z][e]
// (e over the 36 x C x N values, 16 bytes per thread, eight slices in flight).  Layer1 at B = 32 cuts the tiles 85 ways (one 64 x 64
// block per kt: 85 x 3 workgroups fill the 256 slots) = 50 MB of slabs; ONE chain per element — 36 x C x N / 4 = 36 864 threads, 85
// dependent steps each — read them at 0.7 TB/s (138 us); eight chains of eleven run eight times the threads, and the reduce kernel adds
// the eight partial slabs as it reads them.  (Alone the two passes take 18 + 12 us at layer1; beside the data gradient's workgroups,
// which is where a step runs them, 110 + 95 — raised wave priority and non-temporal loads measured equal, scripts/r5/ab_wgrad_passes.sh.)
__global__ __launch_bounds__(256) void conv_wgrad_wino2_sum(float* __restrict__ slab, int S, int per, int64_t n4) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n4) return;
  const int z0 = blockIdx.y * per;
  const int n = min(S, z0 + per) - z0;
  f32x4* q = (f32x4*)slab + (int64_t)z0 * n4 + e;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  int zi = 0;
  for (; zi + 8 <= n; zi += 8) {                                // eight slices in flight, added in slice order
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = q[(zi + u) * n4];
#pragma unroll
    for (int u = 0; u < 8; ++u) a += v[u];
  }
  if (zi < n) {                                                 // the rest of the group in flight together as well
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = zi + u < n ? q[(zi + u) * n4] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (zi + u < n) a += v[u];
  }
  q[0] = a;
}

// how the S tile slices reach the reduce kernel: as they are (S <= 8), or summed in G <= 8 groups of `per` consecutive slices first
static void wino2_wgrad_groups(int S, int* per, int* G) {
  if (S <= 8) { *per = 1; *G = S; return; }
  const int p = (int)slic_cdiv(S, 8) < 8 ? 8 : (int)slic_cdiv(S, 8);
  *per = p;
  *G = (int)slic_cdiv(S, p);
}

static int64_t wino2_tiles(const SlicConvArgs* a) { return (a->M / ((int64_t)a->Hs * a->Ws)) * ((a->Hs + 1) / 2) * ((a->Ws + 3) / 4); }

static void wino2_wgrad_plan(const SlicConvArgs* a, int splits, int* tps, int* S) {
  const int64_t Mt = wino2_tiles(a);
  int64_t per = slic_cdiv(Mt, splits < 1 ? 1 : splits);
  per = slic_cdiv(per, 8) * 8;
  *tps = (int)per;
  *S = (int)slic_cdiv(Mt, per);
}

extern "C" size_t slic_conv_wgrad_wino2_workspace_bytes(const SlicConvArgs* a, int splits) {
  if (!a || a->M <= 0) return 0;
  int tps, S;
  wino2_wgrad_plan(a, splits, &tps, &S);
  return slic_align_up((size_t)S * 3 * 12 * a->Cs * a->N * sizeof(float), 256);
}

extern "C" int slic_conv_wino2_tile_table(const SlicConvArgs* a, uint32_t* tile_tab, void* stream) {
  SLIC_REQUIRE(a && tile_tab && a->M > 0 && a->Ws > 0 && a->Hs > 0 && a->Ga == a->Ts && a->Gb == a->Hs && a->Gc == a->Ws && a->Cs > 0,
               "slic_conv_wino2_tile_table: needs a stride-1 same-size geometry");
  SLIC_REQUIRE(a->M < (1ll << 24), "slic_conv_wino2_tile_table: more than 2^24 output positions (split the batch)");
  conv_wino2_tile_table_kernel<<<dim3((unsigned)slic_cdiv(wino2_tiles(a), 256)), dim3(256), 0, (hipStream_t)stream>>>(*a, (uint2*)tile_tab);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_conv_wgrad_wino2(const SlicConvArgs* a, const float* dy, int splits, const uint32_t* tile_tab, float* dW,
                                     void* workspace, void* stream) {
  SLIC_REQUIRE(a && a->src && dy && dW && workspace && tile_tab && splits >= 1, "slic_conv_wgrad_wino2: bad args");
  SLIC_REQUIRE(a->Cs % 64 == 0 && a->N % 64 == 0 && a->sa == 1 && a->sb == 1 && a->sc == 1 && a->Ga == a->Ts && a->Gb == a->Hs &&
                   a->Gc == a->Ws && !a->k_run_len,
               "slic_conv_wgrad_wino2: needs a 3x3x3 stride-1 same-size geometry, Cs %% 64 == 0, N %% 64 == 0");
  const int64_t dyb = a->M * (int64_t)a->N * 4;
  SLIC_REQUIRE(dyb < (int64_t)0xFFFFFF00u && a->M * (int64_t)a->Cs * 4 < (int64_t)0xFFFFFF00u && a->M < (1ll << 24),
               "slic_conv_wgrad_wino2: tensors larger than 4 GiB / 2^24 positions (split the batch)");
  int tps, S;
  wino2_wgrad_plan(a, splits, &tps, &S);
  hipStream_t st = (hipStream_t)stream;
  constexpr size_t lds = (size_t)WB_STAGES * WB_STAGE_BYTES;
  static_assert(lds <= 160 * 1024, "LDS");
  static bool attr_set = false;
  if (!attr_set) {
    SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)conv_wgrad_wino2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int64_t total = (int64_t)3 * (a->Cs / 64) * (a->N / 64) * S;
  SLIC_REQUIRE(total < (1ll << 30), "slic_conv_wgrad_wino2: grid too large");
  const unsigned gx = (unsigned)((total + 7) / 8 * 8);
  conv_wgrad_wino2_kernel<<<dim3(gx), dim3(512), lds, st>>>(*a, dy, (unsigned)dyb, (const uint2*)tile_tab, (float*)workspace, tps, S);
  SLIC_LAUNCH_CHECK();
#if SLIC_W2_ABL & 4096
  return SLIC_OK;
#endif
  int per, G;
  wino2_wgrad_groups(S, &per, &G);
  const int64_t n4 = (int64_t)36 * a->Cs * a->N / 4;
  if (per > 1) {
    conv_wgrad_wino2_sum<<<dim3((unsigned)slic_cdiv(n4, 256), (unsigned)G), dim3(256), 0, st>>>((float*)workspace, S, per, n4);
    SLIC_LAUNCH_CHECK();
  }
  const dim3 rg((unsigned)(a->Cs / 8), (unsigned)(a->N / 32));
  const int64_t gstride = (int64_t)per * n4 * 4;
  switch (G) {
#define W2_RED(g) case g: conv_wgrad_wino2_reduce<g><<<rg, dim3(256), 0, st>>>((const float*)workspace, gstride, a->Cs, a->N, dW); break;
    W2_RED(1) W2_RED(2) W2_RED(3) W2_RED(4) W2_RED(5) W2_RED(6) W2_RED(7) W2_RED(8)
#undef W2_RED
    default: SLIC_REQUIRE(false, "slic_conv_wgrad_wino2: %d partial slabs", G);
  }
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
