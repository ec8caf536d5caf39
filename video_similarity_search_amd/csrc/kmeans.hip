// k-means on gfx950: E-step on fp32 MFMA, deterministic M-step, sklearn's bookkeeping.
//
// Replaces sklearn.cluster.KMeans as called from /root/reference/clustering/cluster_masks.py:64-71
// (arithmetic: sklearn/cluster/_k_means_lloyd.pyx:168-218, _k_means_common.pyx:167-311,
//  _kmeans.py:174-277).  Floating-point contract shared with oracle/kmeans_oracle.c:
//   score = cnorm[j] - 2 * dot(x_i, c_j),  dot = k-ascending chain of fmaf from +0.
// v_mfma_f32_32x32x2_f32 computes exactly that chain along K (one rounding per product, k = 0
// then k = 1 inside one instruction), so labels are bit-identical to the oracle as long as
// (a) every (point, centroid) pair accumulates in ONE accumulator over the whole D and
// (b) the LDS image hands the MFMA its k's in ascending order.  Both hold below.
//
// Build: -ffp-contract=off (nothing fuses except explicit fmaf/MFMA).
#include "common.h"
#include <type_traits>
#include <math.h>
#include <stdlib.h>

#define KM_BP 128  // points per workgroup (4 waves x 32)
#define KM_BC 128  // centroids per workgroup (4 MFMA row tiles per wave)
#define KM_BK 32   // k per LDS tile

// LDS tile = [128 rows][32 floats]; 16-byte chunk c of row r lives at chunk c ^ ((r>>1)&7):
// the 16 lanes of a ds_read_b128 group (rows distinct mod 16, same chunk) hit 16 different
// 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int km_off(int row, int chunk) {
  return row * KM_BK + ((chunk ^ ((row >> 1) & 7)) << 2);
}

// One workgroup: 128 points x 128 centroids, full D.  Wave w owns points [32w, 32w+32).
// MFMA roles: A = centroid tile (rows i), B = point tile (cols j)  =>  each lane holds ONE point
// (col = lane&31) and 16 centroid rows per accumulator, so the running argmin is in-register.
template <int NCT>
__global__ __launch_bounds__(256) void km_assign_partial(
    const float* __restrict__ X, int64_t N, int D, int ldx, const float* __restrict__ C, int K,
    int ldc, const float* __restrict__ cnorm, float* __restrict__ pscore,
    int32_t* __restrict__ pidx) {
  constexpr int BC = NCT * 32;
  __shared__ __attribute__((aligned(16))) float ldsX[2][KM_BP * KM_BK];
  __shared__ __attribute__((aligned(16))) float ldsC[2][BC * KM_BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t pblock = (int64_t)blockIdx.x * KM_BP;
  const int cblock = blockIdx.y * BC;

  // staging: thread t loads 8 consecutive floats (k0 + 8*(t&3)) of rows (t>>2) and (t>>2)+64
  const int srow = tid >> 2, scp = tid & 3;
  const float* xr[2];
  const float* cr[2];
  bool cvalid[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    int64_t pr = pblock + srow + 64 * p;
    if (pr > N - 1) pr = N - 1;
    xr[p] = X + pr * (int64_t)ldx + scp * 8;
    int c = cblock + srow + 64 * p;
    cvalid[p] = c < K && (srow + 64 * p) < BC;
    if (c > K - 1) c = K - 1;
    cr[p] = C + (int64_t)c * ldc + scp * 8;
  }
  // two register staging sets (tile kt+2 in flight while kt is computed and kt+1 goes to LDS)
  f32x4 gxs[2][2][2], gcs[2][2][2];
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  auto gload = [&](int kt, f32x4 (&gx)[2][2], f32x4 (&gc)[2][2]) {
    const int k0 = kt * KM_BK;
    const bool kin = (k0 + scp * 8) < D;  // D % 8 == 0: a pair of chunks is all in or all out
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if (kin) {
        gx[p][0] = *(const f32x4*)(xr[p] + k0);
        gx[p][1] = *(const f32x4*)(xr[p] + k0 + 4);
      } else {
        gx[p][0] = z4; gx[p][1] = z4;
      }
      if (kin && cvalid[p]) {
        gc[p][0] = *(const f32x4*)(cr[p] + k0);
        gc[p][1] = *(const f32x4*)(cr[p] + k0 + 4);
      } else {
        gc[p][0] = z4; gc[p][1] = z4;
      }
    }
  };
  // de-interleave: chunk 2*scp gets k = 0,2,4,6 of the 8-group, chunk 2*scp+1 gets k = 1,3,5,7,
  // so lane half h reads its four k's (2t+h, t = 0..3) with one ds_read_b128.
  auto lwrite = [&](int buf, f32x4 (&gx)[2][2], f32x4 (&gc)[2][2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = srow + 64 * p;
      f32x4 e = {gx[p][0].x, gx[p][0].z, gx[p][1].x, gx[p][1].z};
      f32x4 o = {gx[p][0].y, gx[p][0].w, gx[p][1].y, gx[p][1].w};
      *(f32x4*)&ldsX[buf][km_off(row, 2 * scp)] = e;
      *(f32x4*)&ldsX[buf][km_off(row, 2 * scp + 1)] = o;
      if (row < BC) {
        f32x4 ce = {gc[p][0].x, gc[p][0].z, gc[p][1].x, gc[p][1].z};
        f32x4 co = {gc[p][0].y, gc[p][0].w, gc[p][1].y, gc[p][1].w};
        *(f32x4*)&ldsC[buf][km_off(row, 2 * scp)] = ce;
        *(f32x4*)&ldsC[buf][km_off(row, 2 * scp + 1)] = co;
      }
    }
  };

  f32x16 acc[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

  const int r = lane & 31, h = lane >> 5;
  const int nk = (D + KM_BK - 1) / KM_BK;
  auto compute = [&](int buf) {
    const float* Cs = ldsC[buf];
    const float* Xs = ldsX[buf];
    f32x4 b[2], a[2][NCT];               // LDS operands of group q+1 are read under group q's MFMAs
    b[0] = *(const f32x4*)&Xs[km_off(32 * wave + r, h)];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) a[0][ct] = *(const f32x4*)&Cs[km_off(32 * ct + r, h)];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cur = q & 1, nxt = cur ^ 1;
      if (q < 3) {
        b[nxt] = *(const f32x4*)&Xs[km_off(32 * wave + r, 2 * (q + 1) + h)];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) a[nxt][ct] = *(const f32x4*)&Cs[km_off(32 * ct + r, 2 * (q + 1) + h)];
      }
      // k order inside every accumulator: q ascending, t ascending, lane half 0 then 1 => k = 8q + 2t + h ascending
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][ct][t], b[cur][t], acc[ct], 0, 0, 0);
      if (q < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1 + NCT, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * NCT, 0);
    }
  };
  gload(0, gxs[0], gcs[0]);
  lwrite(0, gxs[0], gcs[0]);
  if (nk > 1) gload(1, gxs[1], gcs[1]);
  __syncthreads();
  for (int kt0 = 0; kt0 < nk; kt0 += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int kt = kt0 + half;
      if (kt < nk) {
        if (kt + 2 < nk) gload(kt + 2, gxs[half], gcs[half]);
        compute(half);
        if (kt + 1 < nk) lwrite(half ^ 1, gxs[half ^ 1], gcs[half ^ 1]);
        __syncthreads();
      }
    }
  }

  // running argmin over this workgroup's 128 centroids for the lane's point
  float best = INFINITY;
  int bidx = 0x7fffffff;
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int c = cblock + ct * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
      if (c < K) {
        const float s = cnorm[c] - 2.0f * acc[ct][g];
        if (s < best || (s == best && c < bidx)) { best = s; bidx = c; }
      }
    }
  const float ob = __shfl_xor(best, 32);
  const int oi = __shfl_xor(bidx, 32);
  if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
  const int64_t p = pblock + 32 * wave + r;
  if (h == 0 && p < N) {
    pscore[(int64_t)blockIdx.y * N + p] = best;
    pidx[(int64_t)blockIdx.y * N + p] = bidx;
  }
}

// The same E-step with both operand tiles moved HBM -> LDS by the DMA path (buffer_load ... lds; no staging registers,
// no ds_writes, a STAGES-deep ring with counted vmcnt waits, branch-free steady state).  The DMA cannot de-interleave
// k on the fly, so this kernel takes operands whose rows are ALREADY in the LDS image's order: inside every group of
// eight k's, [k0 k2 k4 k6 | k1 k3 k5 k7]  (km_permute_k8: X once per fit, the centres by km_average).  The MFMA then sees
// exactly the operand registers km_assign_partial builds, so scores and labels are bit-identical.
// Tile = BP points x NCT*32 centroids; the four waves are WP x WC (points x centroids), WP = BP / 32, so a wave owns 32
// points and NCT / WC centroid tiles.  <128, 2, 1>: 48 KB LDS, 3 workgroups / CU; <64, 2, 2>: 32 KB, 5 / CU — more,
// shorter workgroups, which is what the N / 128 x K / 64 = 6256-tile grid of the 100k x 512 x 500 case wants
// (6256 / 768 slots = 8.15 rounds -> 9; 12504 / 1280 = 9.8 -> 10).
template <int BP, int NCT, int WC, int STAGES>
__global__ __launch_bounds__(256) void km_assign_dma(
    const float* __restrict__ Xp, int64_t N, int D, int ldx, const float* __restrict__ Cp, int K,
    int ldc, const float* __restrict__ cnorm, float* __restrict__ pscore, int32_t* __restrict__ pidx,
    int32_t* z0, int32_t* z1) {
  extern __shared__ __attribute__((aligned(16))) float km_lds[];
  // the iteration's two device counters (labels changed; M-step workgroups done) start at zero: nothing before this kernel in the
  // iteration touches them, everything after it is ordered behind it on the stream
  if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) {
    if (z0) *z0 = 0;
    if (z1) *z1 = 0;
  }
  constexpr int WP = 4 / WC;
  static_assert(BP == 32 * WP && NCT % WC == 0, "wave grid");
  constexpr int BC = NCT * 32;
  constexpr int TC = NCT / WC;                               // centroid tiles per wave
  constexpr int AL = BP / 32;
  constexpr int STAGE_FLOATS = (BP + BC) * KM_BK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave % WP, wc = wave / WP;
  const int64_t pblock = (int64_t)blockIdx.x * BP;
  const int cblock = blockIdx.y * BC;
  const int srow = tid >> 3;
  const int cq = (tid & 7) ^ ((srow >> 1) & 7);              // SOURCE chunk of this lane (LDS slot = tid & 7)
  // block-local buffer resources: offsets stay 32-bit whatever N is
  const int64_t xrows = (N - pblock) < BP ? (N - pblock) : BP;
  const int crows = (K - cblock) < BC ? (K - cblock) : BC;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(Xp + pblock * (int64_t)ldx), 0, (int)(((xrows - 1) * (int64_t)ldx + D) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(Cp + (int64_t)cblock * ldc), 0, (int)((((int64_t)crows - 1) * ldc + D) * 4), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFF00u;
  unsigned xoff[AL], coff[NCT];
#pragma unroll
  for (int i = 0; i < AL; ++i) xoff[i] = (srow + 32 * i) < xrows ? ((unsigned)(srow + 32 * i) * (unsigned)ldx + cq * 4) * 4u : OOB;
#pragma unroll
  for (int i = 0; i < NCT; ++i) coff[i] = (srow + 32 * i) < crows ? ((unsigned)(srow + 32 * i) * (unsigned)ldc + cq * 4) * 4u : OOB;
  const int klim = D - cq * 4;                                // this lane's chunk of k-tile kt is inside D iff 32 kt < klim
  const int nk = (D + KM_BK - 1) / KM_BK;
  auto issue = [&](int kt, int stage) {
    float* Xs = km_lds + stage * STAGE_FLOATS;
    float* Cs = Xs + BP * KM_BK;
    const bool kin = kt * KM_BK < klim;                       // false for every lane once kt >= nk: all-OOB (zero) DMAs
    const unsigned kb = (unsigned)kt * (KM_BK * 4u);
#pragma unroll
    for (int i = 0; i < AL; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(Xs + (8 * wave + 32 * i) * KM_BK),
                                               16, (int)((kin && xoff[i] != OOB) ? xoff[i] + kb : OOB), 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NCT; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_c, (__attribute__((address_space(3))) void*)(Cs + (8 * wave + 32 * i) * KM_BK),
                                               16, (int)((kin && coff[i] != OOB) ? coff[i] + kb : OOB), 0, 0, 0);
  };
  f32x16 acc[TC];
#pragma unroll
  for (int ct = 0; ct < TC; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
  const int r = lane & 31, h = lane >> 5;
  auto compute = [&](int stage) {
    const float* Xs = km_lds + stage * STAGE_FLOATS;
    const float* Cs = Xs + BP * KM_BK;
    f32x4 b[2], a[2][TC];
    b[0] = *(const f32x4*)&Xs[km_off(32 * wp + r, h)];
#pragma unroll
    for (int ct = 0; ct < TC; ++ct) a[0][ct] = *(const f32x4*)&Cs[km_off(32 * (wc * TC + ct) + r, h)];
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cur = q & 1, nxt = cur ^ 1;
      if (q < 3) {
        b[nxt] = *(const f32x4*)&Xs[km_off(32 * wp + r, 2 * (q + 1) + h)];
#pragma unroll
        for (int ct = 0; ct < TC; ++ct) a[nxt][ct] = *(const f32x4*)&Cs[km_off(32 * (wc * TC + ct) + r, 2 * (q + 1) + h)];
      }
      // k order inside every accumulator: q ascending, t ascending, lane half 0 then 1 => k = 8q + 2t + h ascending
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ct = 0; ct < TC; ++ct)
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][ct][t], b[cur][t], acc[ct], 0, 0, 0);
      if (q < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1 + TC, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * TC, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  constexpr int PER_STAGE = AL + NCT;
#pragma unroll
  for (int t = 0; t < STAGES - 1; ++t) issue(t, t);
  // k-tiles past the end are all-zero DMAs and 0 * 0 MFMAs (exact no-ops on the accumulators): no branch in the loop
  for (int s0 = 0; s0 < nk; s0 += STAGES) {
#pragma unroll
    for (int sidx = 0; sidx < STAGES; ++sidx) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PER_STAGE) : "memory");
      __builtin_amdgcn_s_barrier();
      issue(s0 + sidx + STAGES - 1, (sidx + STAGES - 1) % STAGES);
      compute(sidx);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // the lane's centroid index rises with (ct, g), so a strict '<' alone keeps the first minimum; cn - 2 acc as ONE fma is the
  // same float as the rounded difference (2 acc is exact)
  float best = INFINITY;
  int bidx = 0x7fffffff;
#pragma unroll
  for (int ct = 0; ct < TC; ++ct)
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int c = cblock + (wc * TC + ct) * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
      const float s = c < K ? __builtin_fmaf(-2.0f, acc[ct][g], cnorm[c]) : INFINITY;
      if (s < best) { best = s; bidx = c; }
    }
  const float ob = __shfl_xor(best, 32);
  const int oi = __shfl_xor(bidx, 32);
  if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
  const int64_t p = pblock + 32 * wp + r;
  if (h == 0 && p < N) {
    const int64_t grp = (int64_t)blockIdx.y * WC + wc;       // partial lists are per wave-column of centroids, ascending
    pscore[grp * N + p] = best;
    pidx[grp * N + p] = bidx;
  }
}

// D <= 512, K >= 128: the CENTROID operand lives in registers (the retrieval kernel's layout with the roles of the operands
// swapped, csrc/topk.hip topk_partial_qreg).  A workgroup owns 128 centroids — wave w the 32 of them that form its MFMA A
// operand, NK * 16 registers per lane, loaded once — and streams a contiguous slice of the points through a 4-stage LDS ring:
// the grid is ONE residency round (centroid blocks x point slices <= the CUs), so no launch, prologue or epilogue sits between
// the 16 k-tiles of a 128 x 128 tile as in km_assign_dma (6256 workgroups of 16 k-tiles each at 100k x 512 x 500), the DMA
// of step s + 3 is issued at step s and waited for at step s + 2, and the last quarter of a step fetches the first fragments
// of the next.  Same k order inside every accumulator, same score expression, same tie rule: bit-identical labels.
// Partial lists: ONE per 128-centroid block (blockIdx.y) — the four waves' (score, index) pairs of a finished tile meet in 4 KB of LDS
// behind the ring and wave w combines those of the tile's points 32 w .. 32 w + 31, groups ascending with a strict '<' (what km_combine
// would do with four separate lists: it now reads a quarter of the bytes).
#ifdef KM_STAMPS
__device__ unsigned long long km_cnt[8];
extern "C" int slic_debug_km_counters(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(km_cnt), sizeof(km_cnt));
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(km_cnt), z, sizeof(z)); }
  return 0;
}
static __device__ inline unsigned long long km_now() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
#define KMS(...) __VA_ARGS__
#else
#define KMS(...)
#endif
template <int NK, int ST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void km_assign_creg(
    const float* __restrict__ Xp, int64_t N, int D, int ldx, const float* __restrict__ Cp, int K,
    int ldc, const float* __restrict__ cnorm, float* __restrict__ pscore, int32_t* __restrict__ pidx,
    int32_t* z0, int32_t* z1) {
  extern __shared__ __attribute__((aligned(16))) float km_lds[];
  // the iteration's two device counters (labels changed; M-step workgroups done) start at zero: nothing before this kernel in the
  // iteration touches them, everything after it is ordered behind it on the stream
  if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) {
    if (z0) *z0 = 0;
    if (z1) *z1 = 0;
  }
  static_assert(NK % ST == 0 && ST == 4, "a point tile is a whole number of ring turns; the DMA's LDS base (M0) reaches 64 KB: 4 stages of 16 KB");
  constexpr int BP = 128;
  constexpr int STAGE_FLOATS = BP * KM_BK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  // this workgroup's points: an even split of the ceil(N / 32) 32-point sub-tiles (one MFMA's worth of points) over the
  // gridDim.x slices, walked as 128-point tiles; the LAST tile of a slice may hold fewer than four sub-tiles and then runs a k loop
  // with that many accumulators (100k points over 64 slices: 12 tiles + one sub-tile instead of 13 tiles on the longest slice)
  const int64_t subs = (N + 31) / 32;
  const int64_t u0 = subs * blockIdx.x / gridDim.x, u1 = subs * (blockIdx.x + 1) / gridDim.x;
  const int ntile = (int)((u1 - u0 + 3) / 4);
  const int npt_last = (int)(u1 - u0) - 4 * (ntile - 1);        // 1 .. 4 sub-tiles in the last tile
#ifdef KM_DBG_HOT
  const int64_t pbeg = 0;   // diagnostic: every slice streams the same rows (L2-resident)
#else
  const int64_t pbeg = u0 * 32;
#endif
  const int64_t prow = (N - pbeg) < (u1 - u0) * 32 ? (N - pbeg) : (u1 - u0) * 32;               // rows of the slice
  const int64_t pend = pbeg + prow;
  const int grp = blockIdx.y * 4 + wave;                       // this wave's 32-centroid group
  const int cbase = grp * 32;
  // lane (r, h) of MFMA step (kt, q, t) supplies C[cbase + r][32 kt + 8 q + 4 h + t] (operands are in the k8-permuted order)
  f32x4 cr[NK][4];
  {
    const int crow = cbase + r;
    const float* cp = Cp + (int64_t)(crow < K ? crow : K - 1) * ldc;
#pragma unroll
    for (int kt = 0; kt < NK; ++kt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 32 * kt + 8 * q + 4 * h;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        cr[kt][q] = (c < D && crow < K) ? *(const f32x4*)(cp + c) : z;
      }
  }
  // accumulator element g of this lane is centroid cbase + (g & 3) + 8 (g >> 2) + 4 h
  float cn[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int c = cbase + (g & 3) + 8 * (g >> 2) + 4 * h;
    cn[g] = c < K ? cnorm[c] : INFINITY;
  }
  const int srow = tid >> 3;
  const int cq = (tid & 7) ^ ((srow >> 1) & 7);              // SOURCE chunk of this lane (LDS slot = tid & 7)
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(Xp + pbeg * (int64_t)ldx), 0, prow > 0 ? (int)(((prow - 1) * (int64_t)ldx + D) * 4) : 0, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFF00u;
  unsigned xoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) xoff[i] = ((unsigned)(srow + 32 * i) * (unsigned)ldx + cq * 4) * 4u;
  const int klim = D - cq * 4;
  // ring step (tile, kt): rows past the slice (the ragged last tile, the ring running past the last tile) lie past the buffer
  // resource's range and come back as zeros by themselves; k-tiles past D are sent out of range explicitly
  auto issue = [&](int tile, int kt, int stage) {
    float* Xs = km_lds + stage * STAGE_FLOATS;
    const bool kin = kt * KM_BK < klim;
    const unsigned kb = (unsigned)kt * (KM_BK * 4u);
    const unsigned tb = (unsigned)tile * (unsigned)(BP * ldx * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(Xs + (8 * wave + 32 * i) * KM_BK),
                                               16, (int)(kin ? tb + xoff[i] + kb : OOB), 0, 0, 0);
  };
  KMS(unsigned long long kc_[4] = {0, 0, 0, 0}; const unsigned long long kt00_ = km_now();)
  f32x16 acc[4];                                               // 4 point sub-tiles of 32 x this wave's 32 centroids
  f32x4 b[2][4];                                               // point fragments, double-buffered ACROSS k-tiles
  int64_t op0 = -1;                                            // first point of the finished tile (its lists leave one step later)
  float* xch = km_lds + ST * STAGE_FLOATS;                     // [wave 4][point 128] {score, index} of the finished tile
  // wave w's quarter of the finished tile: min over the four groups, ascending (first minimum wins: lower centroid index), and out
  auto flush = [&]() {
    const int64_t pq = op0 + 32 * wave + r;
    float best = xch[(0 * 128 + 32 * wave + r) * 2];
    int bidx = __float_as_int(xch[(0 * 128 + 32 * wave + r) * 2 + 1]);
#pragma unroll
    for (int sw = 1; sw < 4; ++sw) {
      const float sc = xch[(sw * 128 + 32 * wave + r) * 2];
      const int si = __float_as_int(xch[(sw * 128 + 32 * wave + r) * 2 + 1]);
      if (sc < best) { best = sc; bidx = si; }
    }
    if (h == 0 && pq < pend) { pscore[(int64_t)blockIdx.y * N + pq] = best; pidx[(int64_t)blockIdx.y * N + pq] = bidx; }
  };
#pragma unroll
  for (int u = 0; u < ST - 1; ++u) issue(u / NK, u % NK, u);   // steps 0 .. ST - 2
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (ST - 2)) : "memory");             // step 0 has landed
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) b[0][pt] = *(const f32x4*)&km_lds[km_off(32 * pt + r, h)];
  KMS(kc_[0] = km_now() - kt00_;)
  auto tile_body = [&](const int tile, auto npt_) {
    constexpr int NPT = decltype(npt_)::value;                 // 32-point sub-tiles of this tile
    KMS(const unsigned long long ka_ = km_now();)
#pragma unroll
    for (int pt = 0; pt < 4; ++pt)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[pt][g] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
      // step s = (tile, kt) computes from stage kt % ST.  Outstanding loads here: the DMAs of steps s + 1 .. s + ST - 2 (4 each);
      // stores of the previous tile's results may sit between them — they only make the wait stricter.
      // (lgkmcnt(0): this wave's ds_writes of the previous tile's (score, index) pairs — xch — must have reached LDS before the barrier
      //  that publishes them to the other waves' flush(); gfx950's back-off barrier does not imply it)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * (ST - 3)) : "memory");         // step s + 1 has landed
      __builtin_amdgcn_s_barrier();
      {
        const int kn = kt + ST - 1;                             // step s + ST - 1
        issue(kn >= NK ? tile + 1 : tile, kn >= NK ? kn - NK : kn, kn % ST);
      }
      // the previous tile's partial argmin: combined and written HERE (behind the barrier that makes every wave's pairs visible), a whole
      // k-tile before the next vmcnt wait, not in front of it
      if (kt == 0 && op0 >= 0) flush();
      const float* Xs = km_lds + (kt % ST) * STAGE_FLOATS;
      const float* Xn = km_lds + ((kt + 1) % ST) * STAGE_FLOATS;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cur = q & 1, nxt = cur ^ 1;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)                          // (all four: the step after a tile's last belongs to a full tile)
          b[nxt][pt] = q < 3 ? *(const f32x4*)&Xs[km_off(32 * pt + r, 2 * (q + 1) + h)]
                             : *(const f32x4*)&Xn[km_off(32 * pt + r, h)];        // first fragments of the next step
        // k order inside every accumulator: q ascending, t ascending, lane half 0 then 1 => k ascending (permuted operands)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int pt = 0; pt < NPT; ++pt)
            acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cr[kt][q][t], b[cur][pt][t], acc[pt], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NPT, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    }
    KMS(const unsigned long long kb_ = km_now(); kc_[1] += kb_ - ka_;)
    // argmin over this wave's 32 centroids: ascending index inside the lane, then across the two lane halves; '<' / lower index
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
      // the lane's centroid index rises with g, so a strict '<' alone keeps the first minimum; a centroid past K scores +inf
      // (cn = +inf) and never wins; cn - 2 acc as ONE fma is the same float as the rounded difference (2 acc is exact)
      float best = INFINITY;
      int bidx = 0x7fffffff;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float sc = __builtin_fmaf(-2.0f, acc[pt][g], cn[g]);
        if (sc < best) { best = sc; bidx = cbase + (g & 3) + 8 * (g >> 2) + 4 * h; }
      }
      const float ob = __shfl_xor(best, 32);
      const int oi = __shfl_xor(bidx, 32);
      if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
      if (h == 0) {
        xch[(wave * 128 + 32 * pt + r) * 2] = best;
        xch[(wave * 128 + 32 * pt + r) * 2 + 1] = __int_as_float(bidx);
      }
    }
    op0 = pbeg + (int64_t)tile * BP;
    KMS(kc_[2] += km_now() - kb_;)
  };
  for (int tile = 0; tile < ntile - 1; ++tile) tile_body(tile, std::integral_constant<int, 4>{});
  if (ntile > 0) {
    if (npt_last == 4) tile_body(ntile - 1, std::integral_constant<int, 4>{});
    else if (npt_last == 3) tile_body(ntile - 1, std::integral_constant<int, 3>{});
    else if (npt_last == 2) tile_body(ntile - 1, std::integral_constant<int, 2>{});
    else tile_body(ntile - 1, std::integral_constant<int, 1>{});
  }
  if (op0 >= 0) {
    __syncthreads();
    flush();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the trailing all-zero DMAs must land before the workgroup leaves
  KMS(kc_[3] = km_now() - kt00_; if (lane == 0) for (int i_ = 0; i_ < 4; ++i_) atomicAdd(&km_cnt[i_], kc_[i_]);)
}

// out[i][8g + {0,1,2,3,4,5,6,7}] = in[i][8g + {0,2,4,6,1,3,5,7}]   (D % 8 == 0)
__global__ void km_permute_k8(const float* __restrict__ X, int64_t N, int D8, int ldx, float* __restrict__ Xp, int ldxp) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * D8) return;
  const int64_t i = e / D8;
  const int g = (int)(e % D8);
  const f32x4 v0 = *(const f32x4*)(X + i * ldx + 8 * g), v1 = *(const f32x4*)(X + i * ldx + 8 * g + 4);
  const f32x4 ev = {v0.x, v0.z, v1.x, v1.z}, od = {v0.y, v0.w, v1.y, v1.w};
  *(f32x4*)(Xp + i * ldxp + 8 * g) = ev;
  *(f32x4*)(Xp + i * ldxp + 8 * g + 4) = od;
}

// labels = argmin over the G centroid groups (groups ascending, strict '<' => first index wins).
// HIST (workgroups of KM_SB = 1024 rows): also the per-block label histogram bc[block][K] the M-step's counting sort starts from
// (km_block_hist's output: every entry written) — the labels are in registers here, the separate pass re-read them.
template <bool HIST>
__global__ void km_combine(const float* __restrict__ pscore, const int32_t* __restrict__ pidx,
                           int G, int64_t N, int K, int32_t* __restrict__ labels,
                           const int32_t* __restrict__ labels_old, int32_t* n_changed,
                           float* __restrict__ best_score, int32_t* __restrict__ bc) {
  extern __shared__ int km_hist_c[];
  if constexpr (HIST) {
    for (int j = threadIdx.x; j < K; j += blockDim.x) km_hist_c[j] = 0;
    __syncthreads();
  }
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int changed = 0;
  if (i < N) {
    float best = pscore[i];
    int idx = pidx[i];
    int g = 1;
    for (; g + 4 <= G; g += 4) {                            // four groups' (score, index) loads in flight; compares stay g-ascending
      float sv[4]; int iv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { sv[u] = pscore[(int64_t)(g + u) * N + i]; iv[u] = pidx[(int64_t)(g + u) * N + i]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) if (sv[u] < best) { best = sv[u]; idx = iv[u]; }
    }
    for (; g < G; ++g) {
      const float s = pscore[(int64_t)g * N + i];
      const int ii = pidx[(int64_t)g * N + i];
      if (s < best) { best = s; idx = ii; }
    }
    if (idx < 0 || idx >= K) idx = 0;  // all-NaN row: sklearn's argmin also yields 0
    labels[i] = idx;
    if (best_score) best_score[i] = best;
    if (labels_old) changed = labels_old[i] != idx;
    if constexpr (HIST) atomicAdd(&km_hist_c[idx], 1);      // integer counts: order-free
  }
  if (labels_old) {
    const unsigned long long m = __ballot(changed);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_changed, (int)__popcll(m));
  }
  if constexpr (HIST) {
    __syncthreads();
    for (int j = threadIdx.x; j < K; j += blockDim.x) bc[(int64_t)blockIdx.x * K + j] = km_hist_c[j];
  }
}

__global__ void km_cnorm(const float* __restrict__ C, int K, int D, int ldc,
                         float* __restrict__ cnorm) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= K) return;
  const float* c = C + (int64_t)j * ldc;
  float acc = 0.f;
  int k = 0;
  for (; k + 16 <= D; k += 16) {            // four 16-byte loads in flight; the fmaf chain stays k-ascending
    const f32x4 v0 = *(const f32x4*)(c + k), v1 = *(const f32x4*)(c + k + 4), v2 = *(const f32x4*)(c + k + 8),
                v3 = *(const f32x4*)(c + k + 12);
    acc = fmaf(v0.x, v0.x, acc); acc = fmaf(v0.y, v0.y, acc); acc = fmaf(v0.z, v0.z, acc); acc = fmaf(v0.w, v0.w, acc);
    acc = fmaf(v1.x, v1.x, acc); acc = fmaf(v1.y, v1.y, acc); acc = fmaf(v1.z, v1.z, acc); acc = fmaf(v1.w, v1.w, acc);
    acc = fmaf(v2.x, v2.x, acc); acc = fmaf(v2.y, v2.y, acc); acc = fmaf(v2.z, v2.z, acc); acc = fmaf(v2.w, v2.w, acc);
    acc = fmaf(v3.x, v3.x, acc); acc = fmaf(v3.y, v3.y, acc); acc = fmaf(v3.z, v3.z, acc); acc = fmaf(v3.w, v3.w, acc);
  }
  for (; k < D; ++k) acc = fmaf(c[k], c[k], acc);
  cnorm[j] = acc;
}

// ---------------- M-step: stable counting sort by label, then per-cluster ordered sums --------
#define KM_SB 1024  // rows per sort block (one workgroup)

__global__ __launch_bounds__(KM_SB) void km_block_hist(const int32_t* __restrict__ labels, int64_t N, int K,
                                                       int32_t* __restrict__ bc /* [nblk][K], every entry written */) {
  extern __shared__ int km_hist[];
  for (int j = threadIdx.x; j < K; j += KM_SB) km_hist[j] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * KM_SB + threadIdx.x;
  if (i < N) atomicAdd(&km_hist[labels[i]], 1);        // integer counts: order-free
  __syncthreads();
  for (int j = threadIdx.x; j < K; j += KM_SB) bc[(int64_t)blockIdx.x * K + j] = km_hist[j];
}

// order[off[l] + (rows with label l in earlier blocks) + (rank inside the block)] = row: a stable counting sort.
// Every workgroup first folds the per-block histograms bc[nblk][K] itself (a launch of its own for that scan cost more than the
// nblk * K coalesced, L2-resident loads per workgroup): tot[l] = rows with label l, base[l] = those in earlier blocks; off =
// exclusive scan of tot over clusters in LDS.  Block 0 also writes cnt = tot for the accumulate pass.
// Rank inside the block = (same-label rows in earlier waves) + (same-label lanes below the row in its own wave).  In-wave:
// the wave's distinct labels are peeled one ballot at a time (<= 64 rounds instead of an O(1024) scan per row); across
// waves: per-wave label counts in LDS when they fit (use_wcnt: 16 x K ints), else a scan of the earlier waves' labels.
__global__ __launch_bounds__(KM_SB) void km_place(const int32_t* __restrict__ labels, int64_t N, int K,
                         const int32_t* __restrict__ bc, int nblk, int32_t* __restrict__ cnt,
                         int32_t* __restrict__ order, int use_wcnt) {
  extern __shared__ int km_cl_off[];          // [K] tot -> exclusive scan of tot, [K] base, then [16][K] per-wave label counts
  __shared__ int wtot[KM_SB / 64];
  __shared__ int lab[KM_SB];
  int* base = km_cl_off + K;
  int* wcnt = km_cl_off + 2 * K;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int me = blockIdx.x;
  for (int j = t; j < K; j += KM_SB) {
    int before = 0, tot = 0;
    int b = 0;
    for (; b + 8 <= nblk; b += 8) {
      int v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = bc[(int64_t)(b + u) * K + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) { tot += v[u]; before += (b + u < me) ? v[u] : 0; }
    }
    for (; b < nblk; ++b) {
      const int v = bc[(int64_t)b * K + j];
      tot += v; before += (b < me) ? v : 0;
    }
    km_cl_off[j] = tot;
    base[j] = before;
    if (me == 0) cnt[j] = tot;
  }
  if (use_wcnt)
    for (int j = t; j < (KM_SB / 64) * K; j += KM_SB) wcnt[j] = 0;
  __syncthreads();
  // ---- cluster offsets: thread t owns clusters [t * per, ...); wave shuffle scan of the run totals, then the 16 wave totals
  const int per = (K + KM_SB - 1) / KM_SB;
  int s = 0;
  for (int u = 0; u < per; ++u) {
    const int j = t * per + u;
    if (j < K) s += km_cl_off[j];
  }
  int inc = s;
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(inc, d);
    if (lane >= d) inc += v;
  }
  if (lane == 63) wtot[wave] = inc;
  __syncthreads();
  int run = inc - s;
  for (int w = 0; w < wave; ++w) run += wtot[w];
  for (int u = 0; u < per; ++u) {
    const int j = t * per + u;
    if (j < K) { const int c = km_cl_off[j]; km_cl_off[j] = run; run += c; }
  }
  // ---- ranks
  const int64_t i = (int64_t)blockIdx.x * KM_SB + t;
  const int l = i < N ? labels[i] : -1;
  int intra = 0;
  unsigned long long todo = __ballot(l >= 0);
  while (todo) {                                        // wave-uniform loop over the wave's distinct labels
    const int leader = __ffsll((long long)todo) - 1;
    const int ll = __shfl(l, leader);
    const unsigned long long m = __ballot(l == ll);
    if (l == ll) intra = __popcll(m & ((1ull << lane) - 1ull));
    if (use_wcnt && lane == leader) wcnt[wave * K + ll] = __popcll(m);
    todo &= ~m;
  }
  if (!use_wcnt) lab[t] = l;
  __syncthreads();
  int earlier = 0;
  if (l >= 0) {
    if (use_wcnt) {
      for (int w = 0; w < wave; ++w) earlier += wcnt[w * K + l];
    } else {
      for (int u = 0; u < (wave << 6); ++u) earlier += (lab[u] == l);
    }
  }
  if (i >= N) return;
  order[km_cl_off[l] + base[l] + earlier + intra] = (int32_t)i;
}

// sums[j, 4*c4 .. 4*c4+3] = sequential fp32 sum over the cluster's rows in ascending order.  TOut = double: the same fp32
// sums, stored widened (the payload of the sharded run's fp64 all-reduce).  nch_out (optional, 2 slots): *n_changed split into
// (low 20 bits, the rest) — two numbers whose sums over <= 16 ranks stay exact in fp32 as well.
template <typename TOut>
__global__ __launch_bounds__(128) void km_accumulate(
    const float* __restrict__ X, int D, int ldx, const int32_t* __restrict__ order,
    const int32_t* __restrict__ cnt, TOut* __restrict__ sums, TOut* __restrict__ counts_f,
    const int32_t* __restrict__ n_changed, TOut* __restrict__ nch_out, int xperm) {
  __shared__ int red[128];
  const int j = blockIdx.x;
  const int c4 = blockIdx.y * 128 + threadIdx.x;
  int o = 0;                                           // off[j] = rows of clusters 0..j-1
  for (int u = threadIdx.x; u < j; u += 128) o += cnt[u];
  red[threadIdx.x] = o;
  __syncthreads();
  for (int d = 64; d > 0; d >>= 1) {
    if (threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
    __syncthreads();
  }
  const int n = cnt[j];
  if (blockIdx.y == 0 && threadIdx.x == 0 && counts_f) counts_f[j] = (TOut)n;
  if (j == 0 && blockIdx.y == 0 && threadIdx.x == 0 && nch_out) {
    const int nc = *n_changed;
    nch_out[0] = (TOut)(nc & 0xFFFFF);
    nch_out[1] = (TOut)(nc >> 20);
  }
  if (c4 * 4 >= D) return;
  const int32_t* ord = order + red[0];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int m = 0;
  for (; m + 8 <= n; m += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(X + (int64_t)ord[m + u] * ldx + c4 * 4);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; m < n; ++m) acc += *(const f32x4*)(X + (int64_t)ord[m] * ldx + c4 * 4);
  if (xperm) {
    // X is the E-step's k8-permuted copy (km_permute_k8): this thread's four values are natural columns 8 g + {0, 2, 4, 6} (even
    // chunk) or 8 g + {1, 3, 5, 7} (odd) — the sums are per column, so only the addresses differ
    TOut* d = sums + (int64_t)j * D + (c4 >> 1) * 8 + (c4 & 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) d[2 * i] = (TOut)acc[i];
    return;
  }
  TOut* d = sums + (int64_t)j * D + c4 * 4;
  if constexpr (sizeof(TOut) == 4) {
    *(f32x4*)d = acc;
  } else {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    f64x2 lo = {(double)acc[0], (double)acc[1]}, hi = {(double)acc[2], (double)acc[3]};
    *(f64x2*)d = lo;
    *(f64x2*)(d + 2) = hi;
  }
}

// The M-step's sums for a SMALL shard (N <= 32768 rows, D <= 512) as ONE launch, no counting sort: workgroup j scans the shard's labels
// itself — eight waves, an eighth of the rows each, 16 consecutive labels per lane and step, an exclusive scan of the lanes' match counts
// placing the row ids in ascending order in the wave's LDS list — and then adds cluster j's rows in row order (the order km_place's stable
// sort gives km_accumulate: bit-identical sums), 16 rows in flight.  For the shard one rank holds in a strong-scaled run (12 500 of 100k
// rows over 8 GPUs) km_place + km_accumulate were two launches of 13 and 500 workgroups at 13.7 + 9.1 us: grids too small to be anything but
// latency; this is one launch of K workgroups (scripts/r5/kmeans_small_shard.py).  TOut / nch_out / xperm as km_accumulate.
#define KM_SCAN_MAXN 32768
template <typename TOut>
__global__ __launch_bounds__(512) void km_scan_accumulate(
    const float* __restrict__ X, int N, int D, int ldx, const int32_t* __restrict__ labels, int seg,
    TOut* __restrict__ sums, TOut* __restrict__ counts_f, const int32_t* __restrict__ n_changed, TOut* __restrict__ nch_out, int xperm) {
  extern __shared__ unsigned short km_ids[];                  // [8 waves][seg] row offsets inside the wave's segment
  __shared__ int wcnt[8];
  const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sbeg = w * seg, send = min(N, sbeg + seg);
  unsigned short* my = km_ids + (size_t)w * seg;
  int pos = 0;
  for (int base = sbeg; base < send; base += 64 * 16) {
    const int r0 = base + lane * 16;
    unsigned m = 0;
    if (r0 + 16 <= send) {
      const int4* lp = (const int4*)(labels + r0);           // 64-byte aligned: sbeg and 16-label steps are multiples of 16
      const int4 a = lp[0], b = lp[1], c = lp[2], d = lp[3];
      m = (a.x == j ? 1u : 0u) | (a.y == j ? 2u : 0u) | (a.z == j ? 4u : 0u) | (a.w == j ? 8u : 0u) |
          (b.x == j ? 16u : 0u) | (b.y == j ? 32u : 0u) | (b.z == j ? 64u : 0u) | (b.w == j ? 128u : 0u) |
          (c.x == j ? 256u : 0u) | (c.y == j ? 512u : 0u) | (c.z == j ? 1024u : 0u) | (c.w == j ? 2048u : 0u) |
          (d.x == j ? 4096u : 0u) | (d.y == j ? 8192u : 0u) | (d.z == j ? 16384u : 0u) | (d.w == j ? 32768u : 0u);
    } else {
      for (int u = 0; u < 16; ++u)
        if (r0 + u < send && labels[r0 + u] == j) m |= 1u << u;
    }
    const int c = __popc(m);
    int incl = c;                                              // inclusive scan of the lanes' counts (ascending lanes = ascending rows)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    int p = pos + incl - c;
    while (m) {
      const int u = __ffs(m) - 1;
      my[p++] = (unsigned short)(r0 + u - sbeg);
      m &= m - 1;
    }
    pos += __shfl(incl, 63);
  }
  if (lane == 0) wcnt[w] = pos;
  __syncthreads();
  int n = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) n += wcnt[u];
  if (tid == 0 && counts_f) counts_f[j] = (TOut)n;
  if (j == 0 && tid == 0 && nch_out) {
    const int nc = *n_changed;
    nch_out[0] = (TOut)(nc & 0xFFFFF);
    nch_out[1] = (TOut)(nc >> 20);
  }
  if (tid * 4 >= D) return;                                    // D <= 512: threads 0 .. D / 4 - 1 own four columns each
  const uint32_t rowbytes = (uint32_t)ldx * 4u;
  const char* xc = (const char*)(X + tid * 4);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // the eight lists read as ONE list, rows ascending, 16 rows in flight whatever list they come from (a cluster of 25 rows has ~3 per
  // list: list by list the loads went out one at a time — 17 us for the kernel)
  int off[9];
  off[0] = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) off[u + 1] = off[u] + wcnt[u];
  auto row_of = [&](int q) -> uint32_t {                       // q-th row of the cluster (q < n; the same for every thread)
    int ww = 0;
#pragma unroll
    for (int u = 1; u < 8; ++u) ww += q >= off[u] ? 1 : 0;
    return (uint32_t)(ww * seg) + km_ids[(size_t)ww * seg + (q - off[ww])];
  };
  int q = 0;
  for (; q + 16 <= n; q += 16) {
    f32x4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = *(const f32x4*)(xc + row_of(q + u) * rowbytes);
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u];
  }
  if (q < n) {                                                 // the last, partial batch: its loads in flight together as well
    f32x4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = q + u < n ? *(const f32x4*)(xc + row_of(q + u) * rowbytes) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (q + u < n) acc += v[u];
  }
  if (xperm) {
    TOut* d = sums + (int64_t)j * D + (tid >> 1) * 8 + (tid & 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) d[2 * i] = (TOut)acc[i];
    return;
  }
  TOut* d = sums + (int64_t)j * D + tid * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) d[i] = (TOut)acc[i];
}

__global__ void km_combine_shards(const float* __restrict__ ps, const float* __restrict__ pc,
                                  int64_t stride, int S, int K, int D, float* __restrict__ sums,
                                  float* __restrict__ counts) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t KD = (int64_t)K * D;
  if (e < KD) {
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += ps[(int64_t)s * stride + e];
    sums[e] = a;
  }
  if (e < K) {
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += pc[(int64_t)s * stride + e];
    counts[e] = a;
  }
}

__global__ void km_dist_to_assigned(const float* __restrict__ X, int64_t N, int D, int ldx,
                                    const float* __restrict__ C, int ldc,
                                    const int32_t* __restrict__ labels, float* __restrict__ dist) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float* x = X + i * (int64_t)ldx;
  const float* c = C + (int64_t)labels[i] * ldc;
  float acc = 0.f;
  for (int k = 0; k < D; k += 4) {
    const f32x4 xv = *(const f32x4*)(x + k);
    const f32x4 cv = *(const f32x4*)(c + k);
    float d;
    d = xv.x - cv.x; acc = fmaf(d, d, acc);
    d = xv.y - cv.y; acc = fmaf(d, d, acc);
    d = xv.z - cv.z; acc = fmaf(d, d, acc);
    d = xv.w - cv.w; acc = fmaf(d, d, acc);
  }
  dist[i] = acc;
}

// partial[b] = sum of v[256 b .. 256 b + 255] in double, ascending
__global__ void sum_blocks_f64(const float* __restrict__ v, int64_t N, double* __restrict__ partial) {
  __shared__ float s[256];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  s[threadIdx.x] = i < N ? v[i] : 0.f;
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0;
    for (int u = 0; u < 256; ++u) a += (double)s[u];
    partial[blockIdx.x] = a;
  }
}
__global__ void sum_serial_f64(const double* __restrict__ partial, int64_t n, double* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double a = 0.0;
    for (int64_t u = 0; u < n; ++u) a += partial[u];
    *out = a;
  }
}

// One workgroup: the n_sel farthest rows by (dist desc, row asc); dist is clobbered (-2 marks taken)
__global__ __launch_bounds__(1024) void km_select_far(float* __restrict__ dist, int64_t N, int n_sel,
                                                      int32_t* __restrict__ far_idx,
                                                      float* __restrict__ far_dist) {
  __shared__ float rv[1024];
  __shared__ int64_t ri[1024];
  const int t = threadIdx.x;
  for (int e = 0; e < n_sel; ++e) {
    float bv = -1.0f;
    int64_t bi = N;
    for (int64_t i = t; i < N; i += 1024) {
      const float d = dist[i];
      if (d > bv) { bv = d; bi = i; }  // ascending i per thread: first max kept
    }
    rv[t] = bv; ri[t] = bi;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
      if (t < s) {
        const float ov = rv[t + s];
        const int64_t oi = ri[t + s];
        if (ov > rv[t] || (ov == rv[t] && oi < ri[t])) { rv[t] = ov; ri[t] = oi; }
      }
      __syncthreads();
    }
    if (t == 0) {
      far_idx[e] = (int32_t)ri[0];
      far_dist[e] = rv[0];
      if (ri[0] < N) dist[ri[0]] = -2.0f;
    }
    __syncthreads();
  }
}

// One workgroup, e ascending: sums[old] -= x; sums[new] = x; counts[new] = 1; counts[old] -= 1
__global__ void km_apply_relocation(const float* __restrict__ xfar, int ldf,
                                    const int32_t* __restrict__ old_ids,
                                    const int32_t* __restrict__ new_ids, int n, int D,
                                    float* __restrict__ sums, float* __restrict__ counts) {
  for (int e = 0; e < n; ++e) {
    const int o = old_ids[e], w = new_ids[e];
    const float* x = xfar + (int64_t)e * ldf;
    for (int k = threadIdx.x; k < D; k += blockDim.x) {
      sums[(int64_t)o * D + k] -= x[k];
      sums[(int64_t)w * D + k] = x[k];
    }
    if (threadIdx.x == 0) { counts[w] = 1.0f; counts[o] -= 1.0f; }
    __syncthreads();
  }
}

// _average_centers in sklearn's in-place j-ascending order, _center_shift, and the next E-step's centre norms, one
// workgroup per cluster.  The element-parallel parts (the new row, the squared differences per group of four) are
// computed by all threads; the two order-defining chains run on one lane each from LDS:
//   cnorm_new[j] = k-ascending fmaf chain of c.c                     (same chain as km_cnorm)
//   shift[j]     = sqrt( sum over groups g of (d0^2 + d1^2 + d2^2 + d3^2)_g, g ascending, then the D % 4 tail )
// PARTS (the sharded run, after its one collective): the sums / counts come as W payloads [K*D sums | K counts | ...] `stride`
// elements apart — fp32 rank partials, added here in rank order (the all-gather exchange), or one fp64 row that an all-reduce
// has summed already, rounded here to fp32 (W = 1).  The combined row and count are also written out (out_sums / out_counts:
// the relocation of an empty cluster and km_status read them).
template <typename TIn>
__device__ __forceinline__ float km_comb(const TIn* __restrict__ p, int64_t stride, int W) {
  if constexpr (sizeof(TIn) == 4) {
    float a = 0.f;
    for (int s = 0; s < W; ++s) a += p[(int64_t)s * stride];
    return a;
  } else {
    double a = 0.0;
    for (int s = 0; s < W; ++s) a += p[(int64_t)s * stride];
    return (float)a;
  }
}

// np.argmax(weight_in_clusters) for an empty cluster j (first index of the maximum): the biggest cluster's row is copied,
// averaged already iff it precedes j (sklearn's in-place loop order).  Block-uniform; 128 threads.
template <typename FC>
__device__ __forceinline__ void km_pick_source(FC&& cnt_of, int j, int K, float* mv, int* mi, int& src, float& alpha) {
  const int t = threadIdx.x;
  float bv = -1.f; int bi = 0;
  for (int u = t; u < K; u += 128) { const float c = cnt_of(u); if (c > bv) { bv = c; bi = u; } }
  mv[t] = bv; mi[t] = bi;
  __syncthreads();
  for (int s2 = 64; s2 > 0; s2 >>= 1) {
    if (t < s2) {
      if (mv[t + s2] > mv[t] || (mv[t + s2] == mv[t] && mi[t + s2] < mi[t])) { mv[t] = mv[t + s2]; mi[t] = mi[t + s2]; }
    }
    __syncthreads();
  }
  src = mi[0];
  const float ws = cnt_of(src);
  alpha = (src < j && ws > 0.0f) ? (float)(1.0 / (double)ws) : 1.0f;
}

// row[D] (LDS, written by the caller, not yet synchronised) = the new centre before the optional normalisation: write it out
// (plain + k8-permuted), its squared norm (k-ascending fmaf chain) and the shift against the old centre (group terms, g ascending)
template <bool AGENT_SHIFT = false>
__device__ __forceinline__ void km_average_tail(int j, const float* __restrict__ Co, int D, float* __restrict__ Cn,
                                                float* __restrict__ shift, float* __restrict__ cnorm_new,
                                                float* __restrict__ Cn_perm, int spherical, float* row, float* tg, float* mv) {
  const int t = threadIdx.x;
  __syncthreads();
  if (spherical) {
    // spherical k-means: the new centre is the mean direction, sklearn.preprocessing.normalize(centers) — a zero row stays
    float a = 0.f;
    for (int k = t; k < D; k += 128) a = fmaf(row[k], row[k], a);
    mv[t] = a;
    __syncthreads();
    for (int s2 = 64; s2 > 0; s2 >>= 1) {
      if (t < s2) mv[t] += mv[t + s2];
      __syncthreads();
    }
    const float nrm = sqrtf(mv[0]);
    __syncthreads();
    if (nrm > 0.f)
      for (int k = t; k < D; k += 128) row[k] = row[k] / nrm;
    __syncthreads();
  }
  for (int k = t; k < D; k += 128) {
    const float v = row[k];
    Cn[(int64_t)j * D + k] = v;
    if (Cn_perm) Cn_perm[(int64_t)j * D + (k & ~7) + ((k & 1) << 2) + ((k & 7) >> 1)] = v;   // km_permute_k8 order
  }
  const int ng = D / 4;
  const float* b = Co + (int64_t)j * D;
  for (int g = t; g < ng; g += 128) {
    const int k = 4 * g;
    const float d0 = row[k] - b[k], d1 = row[k + 1] - b[k + 1], d2 = row[k + 2] - b[k + 2], d3 = row[k + 3] - b[k + 3];
    tg[g] = d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
  }
  if (t < D - 4 * ng) { const float d = row[4 * ng + t] - b[4 * ng + t]; tg[ng + t] = d * d; }
  __syncthreads();
  if (t == 0 && cnorm_new) {
    float acc = 0.f;
    for (int k = 0; k < D; ++k) acc = fmaf(row[k], row[k], acc);
    cnorm_new[j] = acc;
  }
  if (t == 64) {
    float r = 0.f;
    const int nt = ng + (D - 4 * ng);
    for (int g = 0; g < nt; ++g) r += tg[g];
    // AGENT_SHIFT: a write-through store, visible device-wide once it has completed (km_accumulate_average's last workgroup
    // reads every shift from another XCD without an L2 write-back fence)
    if constexpr (AGENT_SHIFT) __hip_atomic_store(shift + j, sqrtf(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else shift[j] = sqrtf(r);
  }
}

template <typename TIn, bool PARTS>
__global__ __launch_bounds__(128) void km_average(const TIn* __restrict__ sums, const TIn* __restrict__ counts,
                                                  int64_t stride, int W, float* __restrict__ out_sums,
                                                  float* __restrict__ out_counts,
                                                  const float* __restrict__ Co, int K, int D, float* __restrict__ Cn,
                                                  float* __restrict__ shift, float* __restrict__ cnorm_new,
                                                  float* __restrict__ Cn_perm, int spherical) {
  extern __shared__ float km_avg_lds[];
  float* row = km_avg_lds;                 // [D] the new centre
  float* tg = km_avg_lds + D;              // [D / 4 + (D % 4)] group terms, then tail terms
  __shared__ float mv[128];
  __shared__ int mi[128];
  const int j = blockIdx.x, t = threadIdx.x;
  auto cnt_of = [&](int u) -> float {
    if constexpr (PARTS) return km_comb<TIn>(counts + u, stride, W);
    else return counts[u];
  };
  auto sum_of = [&](int u, int k) -> float {
    if constexpr (PARTS) return km_comb<TIn>(sums + (int64_t)u * D + k, stride, W);
    else return sums[(int64_t)u * D + k];
  };
  const float w = cnt_of(j);
  if constexpr (PARTS) {
    if (t == 0) out_counts[j] = w;
  }
  int src = j;
  float alpha;
  if (w > 0.0f) alpha = (float)(1.0 / (double)w);                                  // the common case needs no argmax
  else km_pick_source(cnt_of, j, K, mv, mi, src, alpha);
  for (int k = t; k < D; k += 128) {
    if constexpr (PARTS) {
      const float own = sum_of(j, k);
      out_sums[(int64_t)j * D + k] = own;
      row[k] = (src == j ? own : sum_of(src, k)) * alpha;
    } else {
      row[k] = sum_of(src, k) * alpha;
    }
  }
  km_average_tail(j, Co, D, Cn, shift, cnorm_new, Cn_perm, spherical, row, tg, mv);
}

// status word: { sum_j shift_j^2, #empty clusters, n_changed, 0 }.  One workgroup; per-thread partials over a fixed
// strided partition, then a fixed binary tree in LDS (deterministic).  n_changed: a device int, or (the sharded run) the
// (low 20 bits, rest) slot pairs of W payloads `stride` elements apart (fp32 or fp64), summed here.
template <typename TIn>
__global__ __launch_bounds__(256) void km_status(const float* __restrict__ shift, const float* __restrict__ counts, int K,
                                                 const int32_t* n_changed, const TIn* __restrict__ nch_parts, int64_t stride,
                                                 int W, double* status) {
  __shared__ double st[256];
  __shared__ int se[256];
  const int t = threadIdx.x;
  double a = 0.0;
  int ne = 0;
  for (int j = t; j < K; j += 256) {
    a += (double)shift[j] * (double)shift[j];
    ne += counts[j] == 0.0f;
  }
  st[t] = a; se[t] = ne;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (t < s2) { st[t] += st[t + s2]; se[t] += se[t + s2]; }
    __syncthreads();
  }
  if (t == 0) {
    status[0] = st[0];
    status[1] = (double)se[0];
    double nc = n_changed ? (double)*n_changed : -1.0;
    if (nch_parts) {
      nc = 0.0;
      for (int s = 0; s < W; ++s) nc += (double)nch_parts[(int64_t)s * stride] + 1048576.0 * (double)nch_parts[(int64_t)s * stride + 1];
    }
    status[2] = nc;
    status[3] = 0.0;
  }
}

// The single-GPU M-step after the counting sort as ONE launch when a 128-thread workgroup spans a whole row (D <= 512):
// km_accumulate<float> (cluster j's ordered fp32 sums), km_average<float, false> on them (an empty cluster sums its source
// cluster's rows itself, in the same order, instead of reading another workgroup's result), and — by the workgroup that
// finishes last — km_status<float>'s word with the same partition and tree (thread t here stands for that kernel's threads t
// and t + 128).  `done`: a device counter that is zero at launch (the E-step kernel resets it each iteration).
__global__ __launch_bounds__(128) void km_accumulate_average(
    const float* __restrict__ X, int D, int ldx, const int32_t* __restrict__ order, const int32_t* __restrict__ cnt,
    float* __restrict__ sums, float* __restrict__ counts_f, const float* __restrict__ Co, int K, float* __restrict__ Cn,
    float* shift, float* __restrict__ cnorm_new, float* __restrict__ Cn_perm, int spherical,
    const int32_t* n_changed, int32_t* done, double* __restrict__ status, int xperm) {
  extern __shared__ float km_avg_lds[];
  float* row = km_avg_lds;
  float* tg = km_avg_lds + D;
  __shared__ float mv[128];
  __shared__ int mi[128];
  __shared__ int red[128];
  __shared__ double st[128];
  __shared__ int is_last;
  const int j = blockIdx.x, t = threadIdx.x;
  auto offset_of = [&](int jj) -> int {                   // rows of clusters 0 .. jj-1 (block-uniform result)
    int o = 0;
    for (int u = t; u < jj; u += 128) o += cnt[u];
    __syncthreads();
    red[t] = o;
    __syncthreads();
    for (int d = 64; d > 0; d >>= 1) {
      if (t < d) red[t] += red[t + d];
      __syncthreads();
    }
    return red[0];
  };
#ifndef KM_ACC_DEPTH
#define KM_ACC_DEPTH 16
#endif
  // rows in flight per batch; the NEXT batch's row ids (wave-uniform: scalar loads) are fetched while this batch's rows are in
  // flight, byte offsets are 32-bit (the launcher checks N * ldx * 4 < 2^32); the adds stay in row order
  const uint32_t rowbytes = (uint32_t)ldx * 4u;
  auto ordered_sum = [&](const int32_t* ord, int n) -> f32x4 {
    constexpr int DP = KM_ACC_DEPTH;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (t * 4 >= D) return acc;
    const char* xc = (const char*)(X + t * 4);
    int m = 0;
    uint32_t nxt[DP];
    if (DP <= n) {
#pragma unroll
      for (int u = 0; u < DP; ++u) nxt[u] = (uint32_t)ord[u];
    }
    for (; m + DP <= n; m += DP) {
      f32x4 v[DP];
#pragma unroll
      for (int u = 0; u < DP; ++u) v[u] = *(const f32x4*)(xc + nxt[u] * rowbytes);
#ifndef KM_ACC_NOPIPE
      if (m + 2 * DP <= n) {
#pragma unroll
        for (int u = 0; u < DP; ++u) nxt[u] = (uint32_t)ord[m + DP + u];
      }
#endif
#pragma unroll
      for (int u = 0; u < DP; ++u) acc += v[u];
#ifdef KM_ACC_NOPIPE
      if (m + 2 * DP <= n) {
#pragma unroll
        for (int u = 0; u < DP; ++u) nxt[u] = (uint32_t)ord[m + DP + u];
      }
#endif
    }
    for (; m < n; ++m) acc += *(const f32x4*)(xc + (uint32_t)ord[m] * rowbytes);
    return acc;
  };
  const int n = cnt[j];
  f32x4 acc = ordered_sum(order + offset_of(j), n);
  if (t == 0) counts_f[j] = (float)n;
  // xperm: X is the E-step's k8-permuted copy (the iteration then streams ONE copy of the data set — 205 MB at 100k x 512, inside the
  // 256 MB Infinity Cache — instead of two): this thread's four values are natural columns cb + {0, 2, 4, 6}
  const int cb = xperm ? (t >> 1) * 8 + (t & 1) : t * 4, cs = xperm ? 2 : 1;
  if (t * 4 < D) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sums[(int64_t)j * D + cb + cs * i] = acc[i];
  }
  int src = j;
  float alpha;
  if (n > 0) {
    alpha = (float)(1.0 / (double)(float)n);
  } else {
    km_pick_source([&](int u) -> float { return (float)cnt[u]; }, j, K, mv, mi, src, alpha);
    acc = ordered_sum(order + offset_of(src), cnt[src]);
  }
  if (t * 4 < D) {
#pragma unroll
    for (int i = 0; i < 4; ++i) row[cb + cs * i] = acc[i] * alpha;
  }
  km_average_tail<true>(j, Co, D, Cn, shift, cnorm_new, Cn_perm, spherical, row, tg, mv);
  // ---- the status word, by the last workgroup to get here.  The only cross-workgroup data is shift[] (cnt and *n_changed come
  // from earlier kernels): thread 64 stored shift[j] write-through, waits for that store to complete, then takes its ticket.
  // (A __threadfence() here instead costs an L2 write-back per workgroup on this multi-XCD part: measured +30 us per launch.)
#ifdef KM_ABL_NOSTATUS
  return;
#endif
  if (t == 64) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    is_last = __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == K - 1;
  }
  __syncthreads();
  if (!is_last) return;
  double a_lo = 0.0, a_hi = 0.0;
  int ne = 0;
  for (int jj = t; jj < K; jj += 256) {
    const double sj = (double)__hip_atomic_load(shift + jj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a_lo += sj * sj;
    ne += cnt[jj] == 0;
  }
  for (int jj = t + 128; jj < K; jj += 256) {
    const double sj = (double)__hip_atomic_load(shift + jj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a_hi += sj * sj;
    ne += cnt[jj] == 0;
  }
  st[t] = a_lo + a_hi; mi[t] = ne;
  __syncthreads();
  for (int s2 = 64; s2 > 0; s2 >>= 1) {
    if (t < s2) { st[t] += st[t + s2]; mi[t] += mi[t + s2]; }
    __syncthreads();
  }
  if (t == 0) {
    status[0] = st[0];
    status[1] = (double)mi[0];
    status[2] = n_changed ? (double)*n_changed : -1.0;
    status[3] = 0.0;
  }
}

// column sums in double: thread = column, blockIdx.y = 1024-row segment
__global__ void col_stats_seg(const float* __restrict__ X, int64_t N, int D, int ldx,
                              double* __restrict__ part /* [nseg][2][D] */) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= D) return;
  const int64_t s = (int64_t)blockIdx.y * 1024;
  const int64_t e = s + 1024 < N ? s + 1024 : N;
  double s1 = 0.0, s2 = 0.0;
  for (int64_t i = s; i < e; ++i) {
    const double v = (double)X[i * (int64_t)ldx + k];
    s1 += v;
    s2 += v * v;
  }
  part[((int64_t)blockIdx.y * 2 + 0) * D + k] = s1;
  part[((int64_t)blockIdx.y * 2 + 1) * D + k] = s2;
}
__global__ void col_stats_fin(const double* __restrict__ part, int nseg, int D,
                              double* __restrict__ cs, double* __restrict__ cq) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= D) return;
  double t1 = 0.0, t2 = 0.0;
  for (int s = 0; s < nseg; ++s) {
    t1 += part[((int64_t)s * 2 + 0) * D + k];
    t2 += part[((int64_t)s * 2 + 1) * D + k];
  }
  cs[k] = t1;
  cq[k] = t2;
}

__global__ void sub_rowvec(const float* __restrict__ X, int64_t N, int D4, int ldx,
                           const float* __restrict__ v, float* __restrict__ out, int ldo) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * D4) return;
  const int64_t i = e / D4;
  const int c = (int)(e - i * D4);
  const f32x4 x = *(const f32x4*)(X + i * ldx + c * 4);
  const f32x4 m = *(const f32x4*)(v + c * 4);
  *(f32x4*)(out + i * ldo + c * 4) = x - m;
}

// one wave per row: ||x|| via double partials (torch.norm accumulates pairwise in fp32; the
// result agrees to <= 1 ulp), then a true division like `data / l2norms`
__global__ void l2norm_rows(const float* __restrict__ X, int64_t N, int D, int ldx,
                            float* __restrict__ out, int ldo) {
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= N) return;
  const int lane = threadIdx.x & 63;
  const float* x = X + row * ldx;
  double s = 0.0;
  for (int k = lane; k < D; k += 64) { const double v = (double)x[k]; s += v * v; }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float nrm = (float)sqrt(s);
  for (int k = lane; k < D; k += 64) out[row * ldo + k] = x[k] / nrm;
}

// k-means++ : squared distances of every row to T candidate rows, min with closest, potentials
// thread = row (x kept in registers is impossible for D=512; stream x once per candidate pair)
#define PP_TMAX 16
__global__ __launch_bounds__(256) void kpp_dist(const float* __restrict__ X, int64_t N, int D,
                                                int ldx, const int32_t* __restrict__ cand, int T,
                                                const float* __restrict__ closest,
                                                float* __restrict__ newdist,
                                                double* __restrict__ bpart /* [nblk][T] */) {
  extern __shared__ float cs[];  // [T][D] candidate rows
  for (int e = threadIdx.x; e < T * D; e += blockDim.x) {
    const int t = e / D, k = e - t * D;
    cs[e] = X[(int64_t)cand[t] * ldx + k];
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float acc[PP_TMAX];
#pragma unroll
  for (int t = 0; t < PP_TMAX; ++t) acc[t] = 0.f;
  if (i < N) {
    const float* x = X + i * (int64_t)ldx;
    for (int k = 0; k < D; k += 4) {
      const f32x4 xv = *(const f32x4*)(x + k);
#pragma unroll
      for (int t = 0; t < PP_TMAX; ++t) {
        if (t < T) {
          const float* c = cs + t * D + k;
          float d;
          d = xv.x - c[0]; acc[t] = fmaf(d, d, acc[t]);
          d = xv.y - c[1]; acc[t] = fmaf(d, d, acc[t]);
          d = xv.z - c[2]; acc[t] = fmaf(d, d, acc[t]);
          d = xv.w - c[3]; acc[t] = fmaf(d, d, acc[t]);
        }
      }
    }
    const float cl = closest ? closest[i] : INFINITY;
#pragma unroll
    for (int t = 0; t < PP_TMAX; ++t)
      if (t < T) {
        acc[t] = fminf(acc[t], cl);
        newdist[(int64_t)t * N + i] = acc[t];
      }
  }
  // block potentials (double, wave shuffle then LDS)
  __shared__ double wp[4][PP_TMAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < PP_TMAX; ++t) {
    if (t < T) {
      double v = i < N ? (double)acc[t] : 0.0;
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) wp[wave][t] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x < T)
    bpart[(int64_t)blockIdx.x * T + threadIdx.x] =
        wp[0][threadIdx.x] + wp[1][threadIdx.x] + wp[2][threadIdx.x] + wp[3][threadIdx.x];
}
// pot[t] = sum over workgroups of bpart[b][t]: one workgroup per candidate, fixed strided partition + fixed tree
__global__ __launch_bounds__(256) void kpp_pot(const double* __restrict__ bpart, int64_t nblk, int T, double* pot) {
  __shared__ double sm[256];
  const int t = blockIdx.x, i = threadIdx.x;
  double a = 0.0;
  for (int64_t b = i; b < nblk; b += 256) a += bpart[b * T + t];
  sm[i] = a;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (i < s2) sm[i] += sm[i + s2];
    __syncthreads();
  }
  if (i == 0) pot[t] = sm[0];
}

// cumsum (double) of v in 1024-element chunks + searchsorted('left')
__global__ void cs_chunk_sums(const float* __restrict__ v, int64_t N, double* __restrict__ csum) {
  __shared__ double w[4];
  const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
  double s = 0.0;
  for (int u = 0; u < 4; ++u) if (i + u < N) s += (double)v[i + u];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) csum[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}
__global__ void cs_scan_chunks(double* csum, int64_t n) {  // inclusive, serial (n ~ N/1024)
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double a = 0.0;
    for (int64_t u = 0; u < n; ++u) { a += csum[u]; csum[u] = a; }
  }
}
// one WAVE per query value: binary search the chunk (inclusive chunk sums), then inside the 1024-element chunk every
// lane sums its 16 consecutive elements, lane 0 scans the 64 lane sums in order, and the lane whose range crosses the
// value walks its 16 elements — searchsorted(cumsum, x, 'left') with the cumsum taken chunk-wise in double
__global__ void cs_search(const float* __restrict__ v, int64_t N, const double* __restrict__ csum,
                          int64_t nchunk, const double* __restrict__ vals, int T,
                          int32_t* __restrict__ idx) {
  const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (t >= T) return;
  const int lane = threadIdx.x & 63;
  const double x = vals[t];
  int64_t lo = 0, hi = nchunk;  // first chunk whose inclusive sum >= x
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (csum[mid] < x) lo = mid + 1; else hi = mid;
  }
  int64_t res = N;
  if (lo < nchunk) {
    const double base = lo ? csum[lo - 1] : 0.0;
    const int64_t s0 = lo * 1024 + (int64_t)lane * 16;
    double part = 0.0;
    for (int u = 0; u < 16; ++u) if (s0 + u < N) part += (double)v[s0 + u];
    // exclusive scan of the lane sums in lane order
    double run = base;
    double mine = 0.0;
    for (int l = 0; l < 64; ++l) {
      const double pl = __shfl(part, l);
      if (l == lane) mine = run;
      run += pl;
    }
    // the crossing lane: first lane whose inclusive prefix reaches x
    const bool crosses = (mine + part >= x) && (mine < x || lane == 0);
    const unsigned long long m = __ballot(mine + part >= x);
    const int first = m ? __ffsll((long long)m) - 1 : -1;
    int64_t r = -1;
    if (lane == first) {
      double a = mine;
      const int64_t e = (lo + 1) * 1024 < N ? (lo + 1) * 1024 : N;
      r = e;
      for (int u = 0; u < 16 && s0 + u < N; ++u) {
        a += (double)v[s0 + u];
        if (a >= x) { r = s0 + u; break; }
      }
    }
    (void)crosses;
    if (first >= 0) res = __shfl(r, first);
    else res = (lo + 1) * 1024 < N ? (lo + 1) * 1024 : N;
  }
  if (res > N - 1) res = N - 1;
  if (lane == 0) idx[t] = (int32_t)res;
}

// ---------------- k-means++ as one device-driven sequence (no host round trip per centre) ----------------------
// Step c: (1) kpp_search: T candidates = searchsorted(cumsum(closest), u[c] * current_pot)  (2) kpp_dist_rows: squared
// distances of every row to the T candidates, min with closest, per-256-row potentials  (3) kpp_select: potentials,
// first-minimum candidate, its index into idx[c], and the inclusive scan of ITS per-256-row sums = the chunked cumsum
// step c+1 searches.  `closest` is never copied: it is row `sel` of the previous step's newdist buffer (double-buffered).

// one wave per row: the row is read as one contiguous run (16 B per lane), candidate rows sit in LDS.  A workgroup covers
// KPP_CH = 64 rows (16 per wave, two rows in flight per wave): ~1600 workgroups at N = 100k, enough waves to hide the
// row loads; its per-candidate sums are the chunk sums kpp_select scans and kpp_search walks.
// The T (<= 16) per-lane partial sums of a row are reduced with a reduce-scatter butterfly: each xor step halves the
// number of values a lane carries (the lane's address bit picks the half it keeps), so a row costs 10 (T <= 8) or 17
// cross-lane moves instead of 6 T; candidate t's total ends up in lanes [8t, 8t+8) (T <= 8) or [4t, 4t+4).
#define KPP_CH 64
template <int TP>
__device__ __forceinline__ float kpp_reduce_scatter(float (&v)[TP], int lane) {
  int off = 32;
#pragma unroll
  for (int half = TP / 2; half >= 1; half >>= 1, off >>= 1) {
    const bool up = (lane & off) != 0;
#pragma unroll
    for (int j = 0; j < half; ++j) {
      const float keep = up ? v[j + half] : v[j];
      const float send = up ? v[j] : v[j + half];
      v[j] = keep + __shfl_xor(send, off);
    }
  }
  float r = v[0];
  for (; off >= 1; off >>= 1) r += __shfl_xor(r, off);
  return r;
}

template <int TP>
__global__ __launch_bounds__(256) void kpp_dist_rows(const float* __restrict__ X, int64_t N, int D, int ldx,
                                                     const int32_t* __restrict__ cand, int T,
                                                     const float* __restrict__ prev, const int32_t* __restrict__ sel,
                                                     float* __restrict__ newdist, double* __restrict__ bpart) {
  extern __shared__ float cs[];  // [T][D] candidate rows
  __shared__ float res[TP][KPP_CH];
  for (int e = threadIdx.x; e < T * D; e += blockDim.x) {
    const int t = e / D, k = e - t * D;
    cs[e] = X[(int64_t)cand[t] * ldx + k];
  }
  for (int e = threadIdx.x; e < TP * KPP_CH; e += blockDim.x) (&res[0][0])[e] = 0.f;
  __syncthreads();
  const float* closest = prev ? prev + (int64_t)(*sel) * N : nullptr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int RPW = KPP_CH / 4;
  constexpr int GL = 64 / TP;                            // lanes per candidate group after the scatter steps
  const int64_t base = (int64_t)blockIdx.x * KPP_CH + wave * RPW;
  for (int rr = 0; rr < RPW; rr += 2) {                  // two rows in flight (four: register pressure costs more than it hides)
    const int64_t i0 = base + rr, i1 = i0 + 1;
    if (i0 >= N) break;                                  // wave-uniform
    const bool two = i1 < N;
    float a0[TP], a1[TP];
#pragma unroll
    for (int t = 0; t < TP; ++t) { a0[t] = 0.f; a1[t] = 0.f; }
    const float* x0 = X + i0 * (int64_t)ldx;
    const float* x1 = X + (two ? i1 : i0) * (int64_t)ldx;
    for (int k0 = lane * 4; k0 < D; k0 += 256) {
      const f32x4 u0 = *(const f32x4*)(x0 + k0);
      const f32x4 u1 = *(const f32x4*)(x1 + k0);
#pragma unroll
      for (int t = 0; t < TP; ++t)
        if (t < T) {
          const f32x4 c = *(const f32x4*)(cs + t * D + k0);
          float d;
          d = u0.x - c.x; a0[t] = fmaf(d, d, a0[t]);
          d = u0.y - c.y; a0[t] = fmaf(d, d, a0[t]);
          d = u0.z - c.z; a0[t] = fmaf(d, d, a0[t]);
          d = u0.w - c.w; a0[t] = fmaf(d, d, a0[t]);
          d = u1.x - c.x; a1[t] = fmaf(d, d, a1[t]);
          d = u1.y - c.y; a1[t] = fmaf(d, d, a1[t]);
          d = u1.z - c.z; a1[t] = fmaf(d, d, a1[t]);
          d = u1.w - c.w; a1[t] = fmaf(d, d, a1[t]);
        }
    }
    const float r0 = kpp_reduce_scatter<TP>(a0, lane);
    const float r1 = kpp_reduce_scatter<TP>(a1, lane);
    if ((lane & (GL - 1)) == 0) {
      res[lane / GL][wave * RPW + rr] = r0;
      if (two) res[lane / GL][wave * RPW + rr + 1] = r1;
    }
  }
  __syncthreads();
  // wave w writes candidates w, w + 4, ...: row r = lane of the 64-row chunk, coalesced; chunk potentials in double
  const int64_t i = (int64_t)blockIdx.x * KPP_CH + lane;
  const float cl = (closest && i < N) ? closest[i] : INFINITY;
  for (int t = wave; t < T; t += 4) {
    const float m = fminf(res[t][lane], cl);
    if (i < N) newdist[(int64_t)t * N + i] = m;
    double v = i < N ? (double)m : 0.0;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) bpart[(int64_t)blockIdx.x * T + t] = v;
  }
}

// The same distances on the matrix pipe: d(x, c) = |x|^2 + |c|^2 - 2 x.c  (sklearn's own formulation,
// euclidean_distances with precomputed row norms), x.c by v_mfma_f32_32x32x2_f32 with A = the T (<= 16) candidate rows
// padded to one 32-row tile and B = 128 points per workgroup, both DMA'd from the k-permuted copy of X (the candidates ARE
// rows of it) through the 2-stage LDS ring of the E-step.  One pass over X per centre at HBM speed instead of a VALU loop.
// A workgroup covers two KPP_CH = 64 row chunks (waves 0-1, waves 2-3).
__global__ __launch_bounds__(256) void kpp_dist_mfma(const float* __restrict__ Xp, const float* __restrict__ xnorm, int64_t N,
                                                     int D, int ldx, const int32_t* __restrict__ cand, int T,
                                                     const float* __restrict__ prev, const int32_t* __restrict__ sel,
                                                     float* __restrict__ newdist, double* __restrict__ bpart, int64_t nchunk) {
  extern __shared__ __attribute__((aligned(16))) float km_lds[];
  constexpr int BP = 128;
  constexpr int STAGE_FLOATS = (BP + 32) * KM_BK;
  __shared__ float cn[PP_TMAX];
  __shared__ double wp[4][PP_TMAX];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t pblock = (int64_t)blockIdx.x * BP;
  const int srow = tid >> 3;
  const int cq = (tid & 7) ^ ((srow >> 1) & 7);
  const int64_t xrows = (N - pblock) < BP ? (N - pblock) : BP;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(Xp + pblock * (int64_t)ldx), 0, (int)(((xrows - 1) * (int64_t)ldx + D) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(
      (void*)Xp, 0, (int)(((N - 1) * (int64_t)ldx + D) * 4), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFF00u;
  unsigned xoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) xoff[i] = (srow + 32 * i) < xrows ? ((unsigned)(srow + 32 * i) * (unsigned)ldx + cq * 4) * 4u : OOB;
  const unsigned coff = srow < T ? ((unsigned)cand[srow] * (unsigned)ldx + cq * 4) * 4u : OOB;     // candidate row `srow`
  if (tid < T) cn[tid] = xnorm[cand[tid]];
  const int klim = D - cq * 4;
  const int nk = (D + KM_BK - 1) / KM_BK;
  auto issue = [&](int kt, int stage) {
    float* Xs = km_lds + stage * STAGE_FLOATS;
    float* Cs = Xs + BP * KM_BK;
    const bool kin = kt * KM_BK < klim;
    const unsigned kb = (unsigned)kt * (KM_BK * 4u);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(Xs + (8 * wave + 32 * i) * KM_BK),
                                               16, (int)((kin && xoff[i] != OOB) ? xoff[i] + kb : OOB), 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_c, (__attribute__((address_space(3))) void*)(Cs + (8 * wave) * KM_BK),
                                             16, (int)((kin && coff != OOB) ? coff + kb : OOB), 0, 0, 0);
  };
  f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  const int r = lane & 31, h = lane >> 5;
  constexpr int PER_STAGE = 5;
  issue(0, 0);
  for (int s0 = 0; s0 < nk; s0 += 2) {
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      issue(s0 + sidx + 1, sidx ^ 1);
      const float* Xs = km_lds + sidx * STAGE_FLOATS;
      const float* Cs = Xs + BP * KM_BK;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b = *(const f32x4*)&Xs[km_off(32 * wave + r, 2 * q + h)];
        const f32x4 a = *(const f32x4*)&Cs[km_off(r, 2 * q + h)];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
      }
    }
  }
  (void)PER_STAGE;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const float* closest = prev ? prev + (int64_t)(*sel) * N : nullptr;
  const int64_t i = pblock + 32 * wave + r;
  const bool iv = i < N;
  const float xn = iv ? xnorm[i] : 0.f;
  const float cl = (closest && iv) ? closest[i] : INFINITY;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int t = (g & 3) + 8 * (g >> 2) + 4 * h;           // candidate row held in accumulator register g of this lane half
    if (t < T) {                                             // uniform per (g, h)
      const float d = fmaxf(xn + cn[t] - 2.0f * acc[g], 0.f);
      const float m = fminf(d, cl);
      if (iv) newdist[(int64_t)t * N + i] = m;
      double v = iv ? (double)m : 0.0;
      for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);     // over the 32 points of this wave (lane half keeps its t)
      if (r == 0) wp[wave][t] = v;
    }
  }
  __syncthreads();
  if (tid < T) {
    const int64_t c0 = (int64_t)blockIdx.x * 2;
    bpart[c0 * T + tid] = wp[0][tid] + wp[1][tid];
    if (c0 + 1 < nchunk) bpart[(c0 + 1) * T + tid] = wp[2][tid] + wp[3][tid];
  }
}

// one workgroup of 16 waves: pot[t] = sum_b bpart[b][t] — wave t sums candidate t's column (lanes strided over the chunks,
// then a fixed shuffle tree) — sel = first minimum, idx_out = cand[sel], csum[b] = inclusive scan over b of bpart[b][sel]
// (per-thread runs, wave shuffle scans, one scan of the 16 wave totals).  Latency-bound: a handful of barriers in all.
#define KPP_ST 1024
// The draw of the NEXT step's candidates (kpp_search's arithmetic, wave t = candidate t) follows in the same launch when
// u_next is given: it needs nothing but this launch's csum / sel / potential, and a separate 7 us launch per centre is a
// tenth of a k-means++ run.  csum and cand are deliberately not __restrict__/const here: they are written and re-read.
__global__ __launch_bounds__(KPP_ST) void kpp_select(const double* __restrict__ bpart, int64_t nblk, int T,
                                                     int32_t* cand, int32_t* __restrict__ sel,
                                                     double* __restrict__ cur_pot, double* csum,
                                                     int32_t* __restrict__ idx_out, const float* __restrict__ nd_cur, int64_t N,
                                                     const double* __restrict__ u_next, int Tn) {
  __shared__ double pots[PP_TMAX];
  __shared__ double wtot[KPP_ST / 64];
  __shared__ int s_sel;
  const int i = threadIdx.x, lane = i & 63, wave = i >> 6;
  if (wave < T) {
    double a = 0.0;
    for (int64_t b = lane; b < nblk; b += 64) a += bpart[b * T + wave];
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) pots[wave] = a;
  }
  __syncthreads();
  if (i == 0) {
    int b = 0;
    for (int t = 1; t < T; ++t) if (pots[t] < pots[b]) b = t;          // np.argmin: first minimum
    s_sel = b;
    *sel = b;
    *cur_pot = pots[b];
    *idx_out = cand[b];
  }
  __syncthreads();
  const int b = s_sel;
  // inclusive scan of column b: thread i owns the run of `per` consecutive chunks [i * per, ...)
  const int64_t per = (nblk + KPP_ST - 1) / KPP_ST;
  const int64_t c0 = (int64_t)i * per, c1 = c0 + per < nblk ? c0 + per : nblk;
  double run = 0.0;
  for (int64_t c = c0; c < c1; ++c) run += bpart[c * T + b];
  double inc = run;                                                     // inclusive scan of the run totals inside the wave
  for (int d = 1; d < 64; d <<= 1) {
    const double u = __shfl_up(inc, d);
    if (lane >= d) inc += u;
  }
  if (lane == 63) wtot[wave] = inc;
  __syncthreads();
  double woff = 0.0;
  for (int w = 0; w < wave; ++w) woff += wtot[w];                       // <= 15 adds, same order in every thread of the wave
  double a = woff + inc - run;                                          // exclusive offset of this thread's run
  for (int64_t c = c0; c < c1; ++c) { a += bpart[c * T + b]; csum[c] = a; }
  if (!u_next) return;
  __syncthreads();                                                       // csum is complete (and visible: one workgroup)
  static_assert(KPP_CH == 64 && KPP_ST / 64 >= PP_TMAX, "one wave per candidate, one element per lane");
  if (wave >= Tn) return;
  // x = u * potential; searchsorted(cumsum(v), x, 'left') with the cumsum taken in KPP_CH-element chunks — kpp_search, verbatim
  const float* v = nd_cur + (int64_t)b * N;
  const double x = u_next[wave] * pots[b];
  int64_t lo = 0, hi = nblk;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (csum[mid] < x) lo = mid + 1; else hi = mid;
  }
  int64_t res = N;
  if (lo < nblk) {
    const double base = lo ? csum[lo - 1] : 0.0;
    const int64_t e = lo * KPP_CH + lane;
    const double part = e < N ? (double)v[e] : 0.0;
    double runs = base, incl = 0.0;
    for (int l = 0; l < 64; ++l) {                       // sequential inclusive scan in lane order
      runs += __shfl(part, l);
      if (l == lane) incl = runs;
    }
    const unsigned long long m = __ballot(incl >= x && e < N);
    res = m ? lo * KPP_CH + (__ffsll((long long)m) - 1) : ((lo + 1) * KPP_CH < N ? (lo + 1) * KPP_CH : N);
  }
  if (res > N - 1) res = N - 1;
  if (lane == 0) cand[wave] = (int32_t)res;
}

// one WAVE per query: x = u * (*cur_pot); searchsorted(cumsum(v), x, 'left') with the cumsum taken in KPP_CH-element
// chunks (inclusive chunk sums csum, one element per lane inside the chunk) in double, clipped to N - 1
__global__ void kpp_search(const float* __restrict__ prev, const int32_t* __restrict__ sel, int64_t N,
                           const double* __restrict__ csum, int64_t nchunk, const double* __restrict__ u,
                           const double* __restrict__ cur_pot, int T, int32_t* __restrict__ idx) {
  static_assert(KPP_CH == 64, "one element per lane");
  const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (t >= T) return;
  const float* v = prev + (int64_t)(*sel) * N;
  const int lane = threadIdx.x & 63;
  const double x = u[t] * (*cur_pot);
  int64_t lo = 0, hi = nchunk;  // first chunk whose inclusive sum >= x
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (csum[mid] < x) lo = mid + 1; else hi = mid;
  }
  int64_t res = N;
  if (lo < nchunk) {
    const double base = lo ? csum[lo - 1] : 0.0;
    const int64_t e = lo * KPP_CH + lane;
    const double part = e < N ? (double)v[e] : 0.0;
    double run = base, incl = 0.0;
    for (int l = 0; l < 64; ++l) {                       // sequential inclusive scan in lane order
      run += __shfl(part, l);
      if (l == lane) incl = run;
    }
    const unsigned long long m = __ballot(incl >= x && e < N);
    res = m ? lo * KPP_CH + (__ffsll((long long)m) - 1) : ((lo + 1) * KPP_CH < N ? (lo + 1) * KPP_CH : N);
  }
  if (res > N - 1) res = N - 1;
  if (lane == 0) idx[t] = (int32_t)res;
}

// ---- the R initialisations of KMeans(n_init = R) side by side ---------------------------------------------------------------
// sklearn runs them one after the other, but the only thing they share is the RNG stream, and k-means++'s draws do not depend on
// the data: with the uniforms drawn up front (in the order the sequential loop draws them) the R runs are independent.  Step c of
// ALL of them is then ONE pass over X — R x T candidate rows as NG tiles of 32 MFMA rows per workgroup instead of one pass per
// run (the single-run pass is HBM-bound at T = 8 rows: 46 us per centre and run) — and one selection launch of R workgroups.
// Row q = r * T + t of the candidate matrix belongs to run r; every run keeps its own closest-distance arrays, potentials,
// chunk sums and picks, with exactly the arithmetic (and order) of the single-run kernels above: the results are bit-identical.
template <int NG>
__global__ __launch_bounds__(256) void kpp_dist_mfma_batch(const float* __restrict__ Xp, const float* __restrict__ xnorm, int64_t N,
                                                           int D, int ldx, const int32_t* __restrict__ cand, int RT, int T, int Tprev,
                                                           const float* __restrict__ prev, const int32_t* __restrict__ sel,
                                                           float* __restrict__ newdist, double* __restrict__ bpart, int64_t nchunk) {
  extern __shared__ __attribute__((aligned(16))) float km_lds[];
  constexpr int BP = 128, CR = 32 * NG;
  constexpr int STAGE_FLOATS = (BP + CR) * KM_BK;
  __shared__ float cn[CR];
  __shared__ double wp[4][CR];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t pblock = (int64_t)blockIdx.x * BP;
  const int srow = tid >> 3;
  const int cq = (tid & 7) ^ ((srow >> 1) & 7);
  const int64_t xrows = (N - pblock) < BP ? (N - pblock) : BP;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(Xp + pblock * (int64_t)ldx), 0, (int)(((xrows - 1) * (int64_t)ldx + D) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(
      (void*)Xp, 0, (int)(((N - 1) * (int64_t)ldx + D) * 4), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFF00u;
  unsigned xoff[4], coff[NG];
#pragma unroll
  for (int i = 0; i < 4; ++i) xoff[i] = (srow + 32 * i) < xrows ? ((unsigned)(srow + 32 * i) * (unsigned)ldx + cq * 4) * 4u : OOB;
#pragma unroll
  for (int g = 0; g < NG; ++g) coff[g] = (srow + 32 * g) < RT ? ((unsigned)cand[srow + 32 * g] * (unsigned)ldx + cq * 4) * 4u : OOB;
  for (int q = tid; q < CR; q += 256) cn[q] = q < RT ? xnorm[cand[q]] : 0.f;
  const int klim = D - cq * 4;
  const int nk = (D + KM_BK - 1) / KM_BK;
  auto issue = [&](int kt, int stage) {
    float* Xs = km_lds + stage * STAGE_FLOATS;
    float* Cs = Xs + BP * KM_BK;
    const bool kin = kt * KM_BK < klim;
    const unsigned kb = (unsigned)kt * (KM_BK * 4u);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(Xs + (8 * wave + 32 * i) * KM_BK),
                                               16, (int)((kin && xoff[i] != OOB) ? xoff[i] + kb : OOB), 0, 0, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_c, (__attribute__((address_space(3))) void*)(Cs + (8 * wave + 32 * g) * KM_BK),
                                               16, (int)((kin && coff[g] != OOB) ? coff[g] + kb : OOB), 0, 0, 0);
  };
  f32x16 acc[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[g][v] = 0.f;
  const int r = lane & 31, h = lane >> 5;
  issue(0, 0);
  for (int s0 = 0; s0 < nk; s0 += 2) {
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      issue(s0 + sidx + 1, sidx ^ 1);
      const float* Xs = km_lds + sidx * STAGE_FLOATS;
      const float* Cs = Xs + BP * KM_BK;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b = *(const f32x4*)&Xs[km_off(32 * wave + r, 2 * q + h)];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const f32x4 a = *(const f32x4*)&Cs[km_off(32 * g + r, 2 * q + h)];
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc[g], 0, 0, 0);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int64_t i = pblock + 32 * wave + r;
  const bool iv = i < N;
  const float xn = iv ? xnorm[i] : 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int q = 32 * g + (e & 3) + 8 * (e >> 2) + 4 * h;      // candidate row held in accumulator element e of this lane half
      if (q < RT) {                                               // uniform per (g, e, h)
        const int run = q / T;
        const float cl = (prev && iv) ? prev[((int64_t)run * Tprev + sel[run]) * N + i] : INFINITY;
        const float d = fmaxf(xn + cn[q] - 2.0f * acc[g][e], 0.f);
        const float m = fminf(d, cl);
        if (iv) newdist[(int64_t)q * N + i] = m;
        double v = iv ? (double)m : 0.0;
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);     // over the 32 points of this wave (the lane half keeps its row)
        if (r == 0) wp[wave][q] = v;
      }
    }
  __syncthreads();
  for (int q = tid; q < RT; q += 256) {
    const int64_t c0 = (int64_t)blockIdx.x * 2;
    bpart[c0 * RT + q] = wp[0][q] + wp[1][q];
    if (c0 + 1 < nchunk) bpart[(c0 + 1) * RT + q] = wp[2][q] + wp[3][q];
  }
}

// kpp_select for run r = blockIdx.x: its T columns of bpart (row stride RT), its sel / potential / chunk sums / candidates / picks
__global__ __launch_bounds__(KPP_ST) void kpp_select_batch(const double* __restrict__ bpart_all, int64_t nblk, int T, int RT, int Tn,
                                                           const int32_t* __restrict__ cand_all, int32_t* __restrict__ cand_next_all,
                                                           int32_t* __restrict__ sel_all,
                                                           double* __restrict__ pot_all, double* csum_all,
                                                           int32_t* __restrict__ idx_out_all, int idx_stride,
                                                           const float* __restrict__ nd_cur_all, int64_t N,
                                                           const double* __restrict__ u_next_all, int64_t u_stride) {
  __shared__ double pots[PP_TMAX];
  __shared__ double wtot[KPP_ST / 64];
  __shared__ int s_sel;
  const int run_id = blockIdx.x;
  const double* bpart = bpart_all + (int64_t)run_id * T;                 // column base; rows are RT apart
  const int32_t* cand = cand_all + (int64_t)run_id * T;                  // this step's candidates (T per run)
  int32_t* cand_next = cand_next_all + (int64_t)run_id * Tn;             // the next step's (Tn per run), in the other buffer: the runs' workgroups are concurrent
  double* csum = csum_all + (int64_t)run_id * nblk;
  const float* nd_cur = nd_cur_all + (int64_t)run_id * T * N;
  const double* u_next = u_next_all ? u_next_all + (int64_t)run_id * u_stride : nullptr;
  const int i = threadIdx.x, lane = i & 63, wave = i >> 6;
  if (wave < T) {
    double a = 0.0;
    for (int64_t b = lane; b < nblk; b += 64) a += bpart[b * RT + wave];
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) pots[wave] = a;
  }
  __syncthreads();
  if (i == 0) {
    int b = 0;
    for (int t = 1; t < T; ++t) if (pots[t] < pots[b]) b = t;          // np.argmin: first minimum
    s_sel = b;
    sel_all[run_id] = b;
    pot_all[run_id] = pots[b];
    idx_out_all[(int64_t)run_id * idx_stride] = cand[b];
  }
  __syncthreads();
  const int b = s_sel;
  const int64_t per = (nblk + KPP_ST - 1) / KPP_ST;
  const int64_t c0 = (int64_t)i * per, c1 = c0 + per < nblk ? c0 + per : nblk;
  double run = 0.0;
  for (int64_t c = c0; c < c1; ++c) run += bpart[c * RT + b];
  double inc = run;
  for (int d = 1; d < 64; d <<= 1) {
    const double u = __shfl_up(inc, d);
    if (lane >= d) inc += u;
  }
  if (lane == 63) wtot[wave] = inc;
  __syncthreads();
  double woff = 0.0;
  for (int w = 0; w < wave; ++w) woff += wtot[w];
  double a = woff + inc - run;
  for (int64_t c = c0; c < c1; ++c) { a += bpart[c * RT + b]; csum[c] = a; }
  if (!u_next) return;
  __syncthreads();
  if (wave >= Tn) return;
  const float* v = nd_cur + (int64_t)b * N;
  const double x = u_next[wave] * pots[b];
  int64_t lo = 0, hi = nblk;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (csum[mid] < x) lo = mid + 1; else hi = mid;
  }
  int64_t res = N;
  if (lo < nblk) {
    const double base = lo ? csum[lo - 1] : 0.0;
    const int64_t e = lo * KPP_CH + lane;
    const double part = e < N ? (double)v[e] : 0.0;
    double runs = base, incl = 0.0;
    for (int l = 0; l < 64; ++l) {
      runs += __shfl(part, l);
      if (l == lane) incl = runs;
    }
    const unsigned long long m = __ballot(incl >= x && e < N);
    res = m ? lo * KPP_CH + (__ffsll((long long)m) - 1) : ((lo + 1) * KPP_CH < N ? (lo + 1) * KPP_CH : N);
  }
  if (res > N - 1) res = N - 1;
  if (lane == 0) cand_next[wave] = (int32_t)res;
}

// ------------------------------------ C ABI ------------------------------------------------
static inline hipStream_t S(void* s) { return (hipStream_t)s; }

extern "C" int slic_kmeans_cnorm(const float* C, int K, int D, int ldc, float* cnorm, void* stream) {
  SLIC_REQUIRE(C && cnorm && K > 0 && D > 0 && ldc >= D, "slic_kmeans_cnorm: bad args");
  km_cnorm<<<dim3((unsigned)slic_cdiv(K, 64)), dim3(64), 0, S(stream)>>>(C, K, D, ldc, cnorm);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

static int km_nct() { return 4; }     // centroid tiles (of 32) per workgroup of the natural-layout E-step (2 was slower)

// quarter tiles (32-point sub-tiles) a workgroup of the centroids-in-registers E-step must have for the kernel to be chosen.  Rounds 2-4: 16
// (four 128-point tiles).  Round 5: 6 — the shard of one rank in an 8-GPU strong-scaled run (12 500 of 100k rows, K = 500: 1.5 tiles per
// workgroup) runs its whole iteration in 125 us with this kernel against 133 with the 128 x 64-tile kernel, 25 000 rows 175 against 186
// (scripts/r5/kmeans_small_shard.py; 4 measured the same as 6).  Same labels either way.  SLIC_KM_CREG_MIN_SUBTILES overrides.
static int km_creg_min_quarter_tiles() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("SLIC_KM_CREG_MIN_SUBTILES");
    v = e ? atoi(e) : 6;
    if (v < 1) v = 1;
  }
  return v;
}

extern "C" size_t slic_kmeans_assign_workspace_bytes(int64_t N, int K) {
  const int64_t G = slic_cdiv(K, 32);          // sized for the finest partial lists (one per 32-centroid wave column)
  return 2 * slic_align_up((size_t)(G * N) * 4, 256);
}

extern "C" int slic_kmeans_assign(const float* X, int64_t N, int D, int ldx, const float* C, int K,
                                  int ldc, const float* cnorm, int32_t* labels,
                                  const int32_t* labels_old, int32_t* n_changed, float* best_score,
                                  void* workspace, void* stream) {
  SLIC_REQUIRE(X && C && cnorm && labels && workspace, "slic_kmeans_assign: null pointer");
  SLIC_REQUIRE(N > 0 && K > 0 && D > 0, "slic_kmeans_assign: N=%lld K=%d D=%d", (long long)N, K, D);
  SLIC_REQUIRE(D % 8 == 0 && ldx % 4 == 0 && ldc % 4 == 0 && ldx >= D && ldc >= D,
               "slic_kmeans_assign: need D %% 8 == 0 and 16-byte aligned rows (D=%d ldx=%d ldc=%d)",
               D, ldx, ldc);
  SLIC_REQUIRE(((uintptr_t)X % 16) == 0 && ((uintptr_t)C % 16) == 0, "slic_kmeans_assign: unaligned");
  SLIC_REQUIRE(!labels_old || n_changed, "slic_kmeans_assign: labels_old needs n_changed");
  const int nct = km_nct();
  const int G = (int)slic_cdiv(K, nct * 32);
  SlicCarver w(workspace);
  float* pscore = w.take<float>((size_t)slic_cdiv(K, 32) * N);
  int32_t* pidx = w.take<int32_t>((size_t)slic_cdiv(K, 32) * N);
  dim3 grid((unsigned)slic_cdiv(N, KM_BP), (unsigned)G);
  if (nct == 2) km_assign_partial<2><<<grid, dim3(256), 0, S(stream)>>>(X, N, D, ldx, C, K, ldc, cnorm, pscore, pidx);
  else km_assign_partial<4><<<grid, dim3(256), 0, S(stream)>>>(X, N, D, ldx, C, K, ldc, cnorm, pscore, pidx);
  SLIC_LAUNCH_CHECK();
  km_combine<false><<<dim3((unsigned)slic_cdiv(N, 256)), dim3(256), 0, S(stream)>>>(
      pscore, pidx, G, N, K, labels, labels_old, n_changed, best_score, nullptr);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

template <int BP, int NCT, int WC, int STAGES>
static int launch_assign_dma(const float* Xp, int64_t N, int D, int ldx, const float* Cp, int K, int ldc,
                             const float* cnorm, float* pscore, int32_t* pidx, hipStream_t st, int32_t* z0 = nullptr,
                             int32_t* z1 = nullptr) {
  const size_t lds = (size_t)STAGES * (BP + NCT * 32) * KM_BK * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)km_assign_dma<BP, NCT, WC, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid((unsigned)slic_cdiv(N, BP), (unsigned)slic_cdiv(K, NCT * 32));
  km_assign_dma<BP, NCT, WC, STAGES><<<grid, dim3(256), lds, st>>>(Xp, N, D, ldx, Cp, K, ldc, cnorm, pscore, pidx, z0, z1);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_kmeans_permute_k8(const float* X, int64_t N, int D, int ldx, float* Xp, int ldxp, void* stream) {
  SLIC_REQUIRE(X && Xp && X != Xp && N > 0 && D > 0 && D % 8 == 0 && ldx % 4 == 0 && ldxp % 4 == 0 && ldx >= D && ldxp >= D,
               "slic_kmeans_permute_k8: need D %% 8 == 0 and 16-byte aligned rows");
  const int64_t tot = N * (D / 8);
  km_permute_k8<<<dim3((unsigned)slic_cdiv(tot, 256)), dim3(256), 0, S(stream)>>>(X, N, D / 8, ldx, Xp, ldxp);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

// bc (optional): also the M-step's per-1024-row-block label histogram [ceil(N / 1024)][K] (km_combine<true>)
static int km_assign_perm_impl(const float* Xp, int64_t N, int D, int ldx, const float* Cp, int K,
                               int ldc, const float* cnorm, int32_t* labels,
                               const int32_t* labels_old, int32_t* n_changed, float* best_score,
                               void* workspace, void* stream, int32_t* bc, int32_t* z0 = nullptr, int32_t* z1 = nullptr) {
  SLIC_REQUIRE(Xp && Cp && cnorm && labels && workspace, "slic_kmeans_assign_perm: null pointer");
  SLIC_REQUIRE(N > 0 && K > 0 && D > 0, "slic_kmeans_assign_perm: N=%lld K=%d D=%d", (long long)N, K, D);
  SLIC_REQUIRE(D % 8 == 0 && ldx % 4 == 0 && ldc % 4 == 0 && ldx >= D && ldc >= D,
               "slic_kmeans_assign_perm: need D %% 8 == 0 and 16-byte aligned rows (D=%d ldx=%d ldc=%d)", D, ldx, ldc);
  SLIC_REQUIRE(((uintptr_t)Xp % 16) == 0 && ((uintptr_t)Cp % 16) == 0, "slic_kmeans_assign_perm: unaligned");
  SLIC_REQUIRE((int64_t)KM_BP * ldx * 4 < (1ll << 31) && (int64_t)128 * ldc * 4 < (1ll << 31), "slic_kmeans_assign_perm: rows too long");
  SLIC_REQUIRE(!labels_old || n_changed, "slic_kmeans_assign_perm: labels_old needs n_changed");
  // tile / ring: 128 points x 64 centroids, 2 stages (measured best: a 64 x 64 tile with 2 x 2 waves fills the grid more evenly
  // but loses the same few % inside the loop; 3 stages and 128 x 128 tiles were slower)
  int G = (int)slic_cdiv(K, 64);
  SlicCarver w(workspace);
  float* pscore = w.take<float>((size_t)slic_cdiv(K, 32) * N);
  int32_t* pidx = w.take<int32_t>((size_t)slic_cdiv(K, 32) * N);
  hipStream_t st = S(stream);
  // Centroids in registers (km_assign_creg) when a 128-centroid block per workgroup wastes little (K = 500 -> 512) and the
  // blocks x slices grid can cover the device: one residency round of 1-workgroup-per-CU workgroups.
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    SLIC_HIP_CHECK(hipGetDevice(&dev));
    SLIC_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  }
  const int ncb = (int)slic_cdiv(K, 128);
  const int64_t tiles = slic_cdiv(N, 128);
  int slices = ncb <= cus ? cus / ncb : 0;
  if (slices > tiles) slices = (int)tiles;
#ifdef KM_NO_CREG
  const bool creg = false &&
#else
  const bool creg = D <= 512 &&
#endif
                     slices >= 1 && (int64_t)ncb * 128 * 7 <= (int64_t)K * 8 &&          // <= 1/8 padding
                    tiles * 4 >= km_creg_min_quarter_tiles() * (int64_t)slices &&             // enough points per workgroup to pay for loading its centroids
                    (slic_cdiv(tiles, slices) * 128 + 128) * (int64_t)ldx * 4 < (1ll << 31);        // slice inside one resource
  if (creg) {
    const size_t lds = (size_t)4 * 128 * KM_BK * sizeof(float) + 4 * 128 * 2 * sizeof(float);     // the ring + the tile's four pair lists
    static bool attr_set = false;
    if (!attr_set) {
      SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)km_assign_creg<16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)km_assign_creg<8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)km_assign_creg<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_set = true;
    }
    dim3 grid((unsigned)slices, (unsigned)ncb);
    if (D > 256) km_assign_creg<16, 4><<<grid, dim3(256), lds, st>>>(Xp, N, D, ldx, Cp, K, ldc, cnorm, pscore, pidx, z0, z1);
    else if (D > 128) km_assign_creg<8, 4><<<grid, dim3(256), lds, st>>>(Xp, N, D, ldx, Cp, K, ldc, cnorm, pscore, pidx, z0, z1);
    else km_assign_creg<4, 4><<<grid, dim3(256), lds, st>>>(Xp, N, D, ldx, Cp, K, ldc, cnorm, pscore, pidx, z0, z1);
    SLIC_LAUNCH_CHECK();
    G = ncb;                                                   // one list per 128-centroid block
  } else {
    int rc = launch_assign_dma<128, 2, 1, 2>(Xp, N, D, ldx, Cp, K, ldc, cnorm, pscore, pidx, st, z0, z1);
    if (rc) return rc;
  }
  if (bc && (size_t)K * 4 <= 48 * 1024)
    km_combine<true><<<dim3((unsigned)slic_cdiv(N, KM_SB)), dim3(KM_SB), (size_t)K * 4, st>>>(pscore, pidx, G, N, K, labels, labels_old,
                                                                                           n_changed, best_score, bc);
  else {
    km_combine<false><<<dim3((unsigned)slic_cdiv(N, 256)), dim3(256), 0, st>>>(pscore, pidx, G, N, K, labels, labels_old, n_changed,
                                                                              best_score, nullptr);
    if (bc) {
      SLIC_LAUNCH_CHECK();
      km_block_hist<<<dim3((unsigned)slic_cdiv(N, KM_SB)), dim3(KM_SB), (size_t)K * 4, st>>>(labels, N, K, bc);
    }
  }
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_kmeans_assign_perm(const float* Xp, int64_t N, int D, int ldx, const float* Cp, int K,
                                       int ldc, const float* cnorm, int32_t* labels,
                                       const int32_t* labels_old, int32_t* n_changed, float* best_score,
                                       void* workspace, void* stream) {
  return km_assign_perm_impl(Xp, N, D, ldx, Cp, K, ldc, cnorm, labels, labels_old, n_changed, best_score, workspace, stream, nullptr);
}

extern "C" size_t slic_kmeans_accumulate_workspace_bytes(int64_t N, int K) {
  const int64_t nblk = slic_cdiv(N, KM_SB);
  return slic_align_up((size_t)nblk * K * 4, 256) + 2 * slic_align_up((size_t)K * 4, 256) +
         slic_align_up((size_t)N * 4, 256);
}

// the per-block histogram slab the accumulate workspace starts with (km_assign_perm_impl can fill it in its combine pass)
static int32_t* km_accumulate_hist_slab(void* workspace) { return (int32_t*)workspace; }
// the shard sizes km_scan_accumulate takes (SLIC_KM_SCAN=0 switches it off: tests compare with the counting-sort path)
static bool km_small_shard(int64_t N, int D, int ldx) {
  const char* e = getenv("SLIC_KM_SCAN");                     // read per call: the tests switch it inside one process
  const bool on = !(e && e[0] == '0');
  return on && N <= KM_SCAN_MAXN && D <= 512 && D % 4 == 0 && (uint64_t)N * (uint64_t)ldx * 4ull < (1ull << 32);
}

// what km_accumulate_average needs beyond the sums: the arguments of slic_kmeans_finalize
struct KmFinish {
  const float* C_old; float* C_new; float* shift; float* cnorm_new; float* C_new_perm; int spherical; double* status;
  bool done;                                     // out: the fused launch ran (else the caller finalises separately)
};
// the M-step's "workgroups done" counter lives in the accumulate workspace's spare K-int slot
static int32_t* km_accumulate_done_counter(void* workspace, int64_t N, int K) {
  return (int32_t*)((char*)workspace + slic_align_up((size_t)slic_cdiv(N, KM_SB) * K * 4, 256) + slic_align_up((size_t)K * 4, 256));
}

template <typename TOut>
static int km_accumulate_impl(const float* X, int64_t N, int D, int ldx, const int32_t* labels, int K, TOut* sums,
                              TOut* counts, const int32_t* n_changed, TOut* nch_out, void* workspace, void* stream,
                              bool have_hist = false, KmFinish* fin = nullptr, int xperm = 0) {
  SLIC_REQUIRE(!xperm || D % 8 == 0, "slic_kmeans_accumulate: a k8-permuted source needs D %% 8 == 0");
  SLIC_REQUIRE(X && labels && sums && counts && workspace, "slic_kmeans_accumulate: null pointer");
  SLIC_REQUIRE(N > 0 && K > 0 && D > 0 && D % 4 == 0 && ldx % 4 == 0 && ldx >= D,
               "slic_kmeans_accumulate: need D %% 4 == 0 (N=%lld K=%d D=%d ldx=%d)", (long long)N, K, D, ldx);
  SLIC_REQUIRE(N < (1ll << 31), "slic_kmeans_accumulate: N too large for int32 row ids");
  const int nblk = (int)slic_cdiv(N, KM_SB);
  SlicCarver w(workspace);
  int32_t* bc = w.take<int32_t>((size_t)nblk * K);
  int32_t* cnt = w.take<int32_t>(K);
  int32_t* done = w.take<int32_t>(K);        // [0]: km_accumulate_average's counter (km_accumulate_done_counter)
  int32_t* order = w.take<int32_t>((size_t)N);
  hipStream_t st = S(stream);
  SLIC_REQUIRE(K <= 16384, "slic_kmeans_accumulate: K > 16384");
  if (km_small_shard(N, D, ldx) && !fin && ((uintptr_t)labels % 16) == 0) {
    // small shard: the workgroups find their rows themselves (no histogram, no counting sort)
    const int seg = (int)slic_cdiv(slic_cdiv(N, 8), 16) * 16;
    const size_t lds = (size_t)8 * seg * sizeof(unsigned short);
    static size_t lds_set = 0;
    if (lds > lds_set) {
      SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)km_scan_accumulate<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(64 * 1024)));
      SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)km_scan_accumulate<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(64 * 1024)));
      lds_set = 64 * 1024;
    }
    km_scan_accumulate<TOut><<<dim3(K), dim3(512), lds, st>>>(X, (int)N, D, ldx, labels, seg, sums, counts, n_changed, nch_out, xperm);
    SLIC_LAUNCH_CHECK();
    return SLIC_OK;
  }
  if (!have_hist) {
    km_block_hist<<<dim3(nblk), dim3(KM_SB), (size_t)K * 4, st>>>(labels, N, K, bc);
    SLIC_LAUNCH_CHECK();
  }
  const int use_wcnt = (size_t)K * 4 * (2 + KM_SB / 64) <= 96 * 1024;
  const size_t lds_place = (size_t)K * 4 * (use_wcnt ? 2 + KM_SB / 64 : 2);
  if (lds_place > 48 * 1024) {
    static size_t lds_set = 0;
    if (lds_place > lds_set) {
      SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)km_place, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_place));
      lds_set = lds_place;
    }
  }
  km_place<<<dim3(nblk), dim3(KM_SB), lds_place, st>>>(labels, N, K, bc, nblk, cnt, order, use_wcnt);
  SLIC_LAUNCH_CHECK();
  if constexpr (sizeof(TOut) == 4) {
    if (fin && D <= 512 && !nch_out && (uint64_t)N * (uint64_t)ldx * 4ull < (1ull << 32)) {
      const size_t lds = ((size_t)D + D / 4 + 4) * sizeof(float);
      km_accumulate_average<<<dim3(K), dim3(128), lds, st>>>(X, D, ldx, order, cnt, sums, counts, fin->C_old, K, fin->C_new, fin->shift,
                                                           fin->cnorm_new, fin->C_new_perm, fin->spherical, n_changed, done,
                                                           fin->status, xperm);
      SLIC_LAUNCH_CHECK();
      fin->done = true;
      return SLIC_OK;
    }
  }
  (void)done;
  km_accumulate<TOut><<<dim3(K, (unsigned)slic_cdiv(D / 4, 128)), dim3(128), 0, st>>>(X, D, ldx, order, cnt, sums, counts,
                                                                                     n_changed, nch_out, xperm);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_kmeans_accumulate(const float* X, int64_t N, int D, int ldx,
                                      const int32_t* labels, int K, float* sums, float* counts,
                                      void* workspace, void* stream) {
  return km_accumulate_impl<float>(X, N, D, ldx, labels, K, sums, counts, nullptr, nullptr, workspace, stream);
}

extern "C" int slic_kmeans_combine_shards(const float* ps, const float* pc, int64_t shard_stride,
                                          int n_shards, int K, int D, float* sums, float* counts,
                                          void* stream) {
  SLIC_REQUIRE(ps && pc && sums && counts && n_shards > 0 && K > 0 && D > 0 && shard_stride > 0,
               "slic_kmeans_combine_shards: bad args");
  const int64_t KD = (int64_t)K * D;
  km_combine_shards<<<dim3((unsigned)slic_cdiv(KD, 256)), dim3(256), 0, S(stream)>>>(ps, pc, shard_stride, n_shards, K, D, sums, counts);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_kmeans_dist_to_assigned(const float* X, int64_t N, int D, int ldx,
                                            const float* C, int ldc, const int32_t* labels,
                                            float* dist, void* stream) {
  SLIC_REQUIRE(X && C && labels && dist && N > 0 && D > 0 && D % 4 == 0 && ldx % 4 == 0 && ldc % 4 == 0,
               "slic_kmeans_dist_to_assigned: bad args (D %% 4 == 0 required)");
  km_dist_to_assigned<<<dim3((unsigned)slic_cdiv(N, 256)), dim3(256), 0, S(stream)>>>(X, N, D, ldx, C, ldc, labels, dist);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" size_t slic_sum_f32_to_f64_workspace_bytes(int64_t N) {
  return slic_align_up((size_t)slic_cdiv(N, 256) * 8, 256);
}
extern "C" int slic_sum_f32_to_f64(const float* v, int64_t N, double* out, void* workspace, void* stream) {
  SLIC_REQUIRE(v && out && workspace && N > 0, "slic_sum_f32_to_f64: bad args");
  const int64_t nb = slic_cdiv(N, 256);
  sum_blocks_f64<<<dim3((unsigned)nb), dim3(256), 0, S(stream)>>>(v, N, (double*)workspace);
  SLIC_LAUNCH_CHECK();
  sum_serial_f64<<<dim3(1), dim3(64), 0, S(stream)>>>((const double*)workspace, nb, out);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_kmeans_select_far(float* dist, int64_t N, int n_sel, int32_t* far_idx,
                                      float* far_dist, void* stream) {
  SLIC_REQUIRE(dist && far_idx && far_dist && N > 0 && n_sel > 0, "slic_kmeans_select_far: bad args");
  km_select_far<<<dim3(1), dim3(1024), 0, S(stream)>>>(dist, N, n_sel, far_idx, far_dist);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_kmeans_apply_relocation(const float* xfar, int ldf, const int32_t* old_ids,
                                            const int32_t* new_ids, int n, int D, float* sums,
                                            float* counts, void* stream) {
  SLIC_REQUIRE(xfar && old_ids && new_ids && sums && counts && n > 0 && D > 0 && ldf >= D,
               "slic_kmeans_apply_relocation: bad args");
  km_apply_relocation<<<dim3(1), dim3(256), 0, S(stream)>>>(xfar, ldf, old_ids, new_ids, n, D, sums, counts);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_kmeans_finalize(const float* C_old, const float* sums, const float* counts,
                                    int K, int D, float* C_new, float* shift, float* cnorm_new,
                                    float* C_new_perm, int spherical, const int32_t* n_changed, double* status,
                                    void* stream) {
  SLIC_REQUIRE(C_old && sums && counts && C_new && shift && status && K > 0 && D > 0,
               "slic_kmeans_finalize: bad args");
  SLIC_REQUIRE(C_new != sums && C_new != C_old, "slic_kmeans_finalize: C_new must not alias");
  SLIC_REQUIRE(D <= 8192, "slic_kmeans_finalize: D > 8192");
  SLIC_REQUIRE(!C_new_perm || D % 8 == 0, "slic_kmeans_finalize: C_new_perm needs D %% 8 == 0");
  hipStream_t st = S(stream);
  const size_t lds = ((size_t)D + D / 4 + 4) * sizeof(float);
  km_average<float, false><<<dim3(K), dim3(128), lds, st>>>(sums, counts, 0, 1, nullptr, nullptr, C_old, K, D, C_new, shift,
                                                            cnorm_new, C_new_perm, spherical);
  SLIC_LAUNCH_CHECK();
  km_status<float><<<dim3(1), dim3(256), 0, st>>>(shift, counts, K, n_changed, nullptr, 0, 0, status);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

// One whole single-GPU Lloyd iteration in one call (the host loop then costs one foreign call per iteration):
// zero n_changed, E-step on the permuted operands, ordered M-step sums, averaging + shift + next norms + status word.
extern "C" size_t slic_kmeans_lloyd_step_workspace_bytes(int64_t N, int K) {
  return slic_align_up(slic_kmeans_assign_workspace_bytes(N, K), 256) + slic_kmeans_accumulate_workspace_bytes(N, K);
}

extern "C" int slic_kmeans_lloyd_step(const float* X, const float* Xp, int64_t N, int D, int ldx,
                                      const float* C_old, const float* Cp_old, const float* cnorm_old, int K,
                                      int32_t* labels, const int32_t* labels_old, int32_t* n_changed,
                                      float* sums, float* counts, float* C_new, float* Cp_new, float* cnorm_new,
                                      float* shift, int spherical, double* status, void* workspace, void* stream) {
  SLIC_REQUIRE(X && Xp && C_old && Cp_old && cnorm_old && labels && n_changed && sums && counts && C_new && Cp_new &&
               cnorm_new && shift && status && workspace, "slic_kmeans_lloyd_step: null pointer");
  SLIC_REQUIRE(C_new != sums && C_new != C_old && D <= 8192 && D % 8 == 0, "slic_kmeans_lloyd_step: C_new must not alias; D %% 8 == 0, D <= 8192");
  char* ws = (char*)workspace;
  void* ws2 = ws + slic_align_up(slic_kmeans_assign_workspace_bytes(N, K), 256);
  // the E-step kernel zeroes *n_changed and the M-step's done counter (no memset launch)
  int rc = km_assign_perm_impl(Xp, N, D, ldx, Cp_old, K, D, cnorm_old, labels, labels_old, n_changed, nullptr, ws, stream,
                               km_accumulate_hist_slab(ws2), n_changed, km_accumulate_done_counter(ws2, N, K));
  if (rc) return rc;
  KmFinish fin = {C_old, C_new, shift, cnorm_new, Cp_new, spherical, status, false};
  // the M-step gathers its rows from the PERMUTED copy too (column sums do not care; the kernels write them back in natural order)
  rc = km_accumulate_impl<float>(Xp, N, D, ldx, labels, K, sums, counts, n_changed, nullptr, ws2, stream, true, &fin, 1);
  if (rc || fin.done) return rc;
  return slic_kmeans_finalize(C_old, sums, counts, K, D, C_new, shift, cnorm_new, Cp_new, spherical, n_changed, status, stream);
}

// The sharded iteration in two calls around its ONE collective:
//   local : *n_changed = 0; E-step on this rank's rows; ordered M-step sums into the payload
//           [K*D sums | K counts | n_changed low 20 bits | n_changed >> 20]  (fp32, or the same numbers widened to fp64)
//   (host) all-gather of the fp32 payloads, or all-reduce(sum) of the fp64 payload, over RCCL
//   global: combine + _average_centers + shift + next norms + permuted centres + the status word
extern "C" size_t slic_kmeans_lloyd_local_workspace_bytes(int64_t N, int K) {
  return slic_kmeans_lloyd_step_workspace_bytes(N, K) + 256;
}

extern "C" int slic_kmeans_lloyd_local(const float* X, const float* Xp, int64_t N, int D, int ldx, const float* Cp_old,
                                       const float* cnorm_old, int K, int32_t* labels, const int32_t* labels_old,
                                       void* payload, int payload_f64, void* workspace, void* stream) {
  SLIC_REQUIRE(X && Xp && Cp_old && cnorm_old && labels && payload && workspace, "slic_kmeans_lloyd_local: null pointer");
  char* ws = (char*)workspace;
  const size_t a1 = slic_align_up(slic_kmeans_assign_workspace_bytes(N, K), 256);
  void* ws2 = ws + a1;
  int32_t* n_changed = (int32_t*)(ws + slic_kmeans_lloyd_step_workspace_bytes(N, K));
  int rc = km_assign_perm_impl(Xp, N, D, ldx, Cp_old, K, D, cnorm_old, labels, labels_old, n_changed, nullptr, ws, stream,
                               km_small_shard(N, D, ldx) ? nullptr : km_accumulate_hist_slab(ws2), n_changed);
  if (rc) return rc;
  const int64_t KD = (int64_t)K * D;
  if (payload_f64) {
    double* p = (double*)payload;
    return km_accumulate_impl<double>(Xp, N, D, ldx, labels, K, p, p + KD, n_changed, p + KD + K, ws2, stream, !km_small_shard(N, D, ldx), nullptr, 1);
  }
  float* p = (float*)payload;
  return km_accumulate_impl<float>(Xp, N, D, ldx, labels, K, p, p + KD, n_changed, p + KD + K, ws2, stream, !km_small_shard(N, D, ldx), nullptr, 1);
}

extern "C" int slic_kmeans_lloyd_global(const void* parts, int parts_f64, int64_t stride, int n_parts, const float* C_old,
                                        int K, int D, float* sums, float* counts, float* C_new, float* Cp_new,
                                        float* cnorm_new, float* shift, int spherical, double* status, void* stream) {
  SLIC_REQUIRE(parts && C_old && sums && counts && C_new && shift && status && K > 0 && D > 0 && n_parts > 0,
               "slic_kmeans_lloyd_global: bad args");
  const int64_t KD = (int64_t)K * D;
  SLIC_REQUIRE(stride >= KD + K + 2 || n_parts == 1, "slic_kmeans_lloyd_global: payload stride %lld < K*D + K + 2", (long long)stride);
  SLIC_REQUIRE(C_new != sums && C_new != C_old, "slic_kmeans_lloyd_global: C_new must not alias");
  SLIC_REQUIRE(D <= 8192, "slic_kmeans_lloyd_global: D > 8192");
  SLIC_REQUIRE(!Cp_new || D % 8 == 0, "slic_kmeans_lloyd_global: Cp_new needs D %% 8 == 0");
  hipStream_t st = S(stream);
  const size_t lds = ((size_t)D + D / 4 + 4) * sizeof(float);
  if (parts_f64) {
    const double* p = (const double*)parts;
    km_average<double, true><<<dim3(K), dim3(128), lds, st>>>(p, p + KD, stride, n_parts, sums, counts, C_old, K, D, C_new, shift,
                                                              cnorm_new, Cp_new, spherical);
    SLIC_LAUNCH_CHECK();
    km_status<double><<<dim3(1), dim3(256), 0, st>>>(shift, counts, K, nullptr, p + KD + K, stride, n_parts, status);
  } else {
    const float* p = (const float*)parts;
    km_average<float, true><<<dim3(K), dim3(128), lds, st>>>(p, p + KD, stride, n_parts, sums, counts, C_old, K, D, C_new, shift,
                                                             cnorm_new, Cp_new, spherical);
    SLIC_LAUNCH_CHECK();
    km_status<float><<<dim3(1), dim3(256), 0, st>>>(shift, counts, K, nullptr, p + KD + K, stride, n_parts, status);
  }
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" size_t slic_col_stats_workspace_bytes(int64_t N, int D) {
  return slic_align_up((size_t)slic_cdiv(N, 1024) * 2 * D * 8, 256);
}
extern "C" int slic_col_stats(const float* X, int64_t N, int D, int ldx, double* col_sum,
                              double* col_sumsq, void* workspace, void* stream) {
  SLIC_REQUIRE(X && col_sum && col_sumsq && workspace && N > 0 && D > 0 && ldx >= D, "slic_col_stats: bad args");
  const int nseg = (int)slic_cdiv(N, 1024);
  col_stats_seg<<<dim3((unsigned)slic_cdiv(D, 64), nseg), dim3(64), 0, S(stream)>>>(X, N, D, ldx, (double*)workspace);
  SLIC_LAUNCH_CHECK();
  col_stats_fin<<<dim3((unsigned)slic_cdiv(D, 64)), dim3(64), 0, S(stream)>>>((const double*)workspace, nseg, D, col_sum, col_sumsq);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_sub_rowvec(const float* X, int64_t N, int D, int ldx, const float* v, float* out,
                               int ldo, void* stream) {
  SLIC_REQUIRE(X && v && out && N > 0 && D > 0 && D % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0,
               "slic_sub_rowvec: bad args (D %% 4 == 0 required)");
  const int64_t tot = N * (D / 4);
  sub_rowvec<<<dim3((unsigned)slic_cdiv(tot, 256)), dim3(256), 0, S(stream)>>>(X, N, D / 4, ldx, v, out, ldo);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_l2norm_rows(const float* X, int64_t N, int D, int ldx, float* out, int ldo, void* stream) {
  SLIC_REQUIRE(X && out && N > 0 && D > 0 && ldx >= D && ldo >= D, "slic_l2norm_rows: bad args");
  l2norm_rows<<<dim3((unsigned)slic_cdiv(N, 4)), dim3(256), 0, S(stream)>>>(X, N, D, ldx, out, ldo);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" size_t slic_kmeanspp_step_workspace_bytes(int64_t N, int T) {
  return slic_align_up((size_t)slic_cdiv(N, 256) * T * 8, 256);
}
extern "C" int slic_kmeanspp_step(const float* X, int64_t N, int D, int ldx, const int32_t* cand,
                                  int T, const float* closest, float* newdist, double* pot,
                                  void* workspace, void* stream) {
  SLIC_REQUIRE(X && cand && newdist && pot && workspace, "slic_kmeanspp_step: null pointer");
  SLIC_REQUIRE(T >= 1 && T <= PP_TMAX && D % 4 == 0 && ldx % 4 == 0 && (size_t)T * D * 4 <= 64 * 1024,
               "slic_kmeanspp_step: need 1 <= T <= %d, D %% 4 == 0, T*D*4 <= 64 KiB", PP_TMAX);
  const int64_t nblk = slic_cdiv(N, 256);
  kpp_dist<<<dim3((unsigned)nblk), dim3(256), (size_t)T * D * 4, S(stream)>>>(X, N, D, ldx, cand, T, closest, newdist, (double*)workspace);
  SLIC_LAUNCH_CHECK();
  kpp_pot<<<dim3(T), dim3(256), 0, S(stream)>>>((const double*)workspace, nblk, T, pot);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

// workspace: newdist [2][T][N] f32 | bpart [nblk][T] f64 | csum [nblk] f64 | cur_pot f64 | cand [T] i32 | sel i32
extern "C" size_t slic_kmeanspp_run_workspace_bytes(int64_t N, int T) {
  const int64_t nblk = slic_cdiv(N, KPP_CH);
  return slic_align_up((size_t)2 * T * N * 4, 256) + slic_align_up((size_t)nblk * T * 8, 256) +
         slic_align_up((size_t)nblk * 8, 256) + 256 + slic_align_up((size_t)T * 4, 256) + 256;
}

extern "C" int slic_kmeanspp_run(const float* X, int64_t N, int D, int ldx, int first, int K, int T,
                                 const double* uniforms, int32_t* idx_out, const float* Xp, const float* xnorm,
                                 void* workspace, void* stream) {
  SLIC_REQUIRE(X && uniforms && idx_out && workspace, "slic_kmeanspp_run: null pointer");
  SLIC_REQUIRE(N > 0 && K > 0 && K <= N && first >= 0 && first < N && T >= 1 && T <= PP_TMAX && D % 4 == 0 && ldx % 4 == 0 &&
               (size_t)T * D * 4 <= 48 * 1024, "slic_kmeanspp_run: need 1 <= T <= %d, D %% 4 == 0, T*D*4 <= 48 KiB", PP_TMAX);
  hipStream_t st = S(stream);
  const int64_t nblk = slic_cdiv(N, KPP_CH);
  SlicCarver w(workspace);
  float* nd = w.take<float>((size_t)2 * T * N);
  double* bpart = w.take<double>((size_t)nblk * T);
  double* csum = w.take<double>((size_t)nblk);
  double* cur_pot = w.take<double>(1);
  int32_t* cand = w.take<int32_t>(T);
  int32_t* sel = w.take<int32_t>(1);
  // matrix-pipe distances when the caller has the k-permuted copy and the row norms (and 32-bit offsets reach every row)
  const bool mfma = Xp && xnorm && D % 8 == 0 && (int64_t)N * ldx * 4 < (1ll << 31);
  const size_t lds_m = (size_t)2 * (128 + 32) * KM_BK * sizeof(float);
  if (mfma) {
    static bool attr_set = false;
    if (!attr_set) {
      SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)kpp_dist_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m));
      attr_set = true;
    }
  }
  const unsigned nblk_m = (unsigned)slic_cdiv(N, 128);
  // centre 0: one candidate (the uniformly drawn row), no closest yet
  SLIC_HIP_CHECK(hipMemcpyAsync(cand, &first, sizeof(int32_t), hipMemcpyHostToDevice, st));
  if (mfma) kpp_dist_mfma<<<dim3(nblk_m), dim3(256), lds_m, st>>>(Xp, xnorm, N, D, ldx, cand, 1, nullptr, nullptr, nd, bpart, nblk);
  else kpp_dist_rows<8><<<dim3((unsigned)nblk), dim3(256), (size_t)D * 4, st>>>(X, N, D, ldx, cand, 1, nullptr, nullptr, nd, bpart);
  SLIC_LAUNCH_CHECK();
  // every select also draws the next step's T candidates (uniforms of step c at uniforms + (c - 1) T) from the distances it chose
  kpp_select<<<dim3(1), dim3(KPP_ST), 0, st>>>(bpart, nblk, 1, cand, sel, cur_pot, csum, idx_out, nd, N, K > 1 ? uniforms : nullptr, T);
  SLIC_LAUNCH_CHECK();
  for (int c = 1; c < K; ++c) {
    const float* prev = nd + (size_t)((c - 1) & 1) * T * N;
    float* cur = nd + (size_t)(c & 1) * T * N;
    if (mfma) kpp_dist_mfma<<<dim3(nblk_m), dim3(256), lds_m, st>>>(Xp, xnorm, N, D, ldx, cand, T, prev, sel, cur, bpart, nblk);
    else if (T <= 8) kpp_dist_rows<8><<<dim3((unsigned)nblk), dim3(256), (size_t)T * D * 4, st>>>(X, N, D, ldx, cand, T, prev, sel, cur, bpart);
    else kpp_dist_rows<16><<<dim3((unsigned)nblk), dim3(256), (size_t)T * D * 4, st>>>(X, N, D, ldx, cand, T, prev, sel, cur, bpart);
    SLIC_LAUNCH_CHECK();
    kpp_select<<<dim3(1), dim3(KPP_ST), 0, st>>>(bpart, nblk, T, cand, sel, cur_pot, csum, idx_out + c, cur, N,
                                                 c + 1 < K ? uniforms + (size_t)c * T : nullptr, T);
    SLIC_LAUNCH_CHECK();
  }
  return SLIC_OK;
}

// All R initialisations of KMeans(n_init = R) in lock-step (kpp_dist_mfma_batch / kpp_select_batch above): firsts[r] = run r's
// uniformly drawn first row, uniforms [R][K - 1][T] (host-drawn in the sequential loop's order), idx_out [R][K].
// Needs the k-permuted copy and the row norms (the matrix-pipe distance kernel); R * T <= 160.
extern "C" size_t slic_kmeanspp_run_batch_workspace_bytes(int64_t N, int T, int R) {
  const int64_t nblk = slic_cdiv(N, KPP_CH);
  return slic_align_up((size_t)2 * R * T * N * 4, 256) + slic_align_up((size_t)nblk * R * T * 8, 256) +
         slic_align_up((size_t)R * nblk * 8, 256) + slic_align_up((size_t)R * 8, 256) + 2 * slic_align_up((size_t)R * T * 4, 256) +
         slic_align_up((size_t)R * 4, 256);
}

extern "C" int slic_kmeanspp_run_batch(const float* Xp, const float* xnorm, int64_t N, int D, int ldx, int R, const int32_t* firsts,
                                       int K, int T, const double* uniforms, int32_t* idx_out, void* workspace, void* stream) {
  SLIC_REQUIRE(Xp && xnorm && firsts && uniforms && idx_out && workspace, "slic_kmeanspp_run_batch: null pointer");
  SLIC_REQUIRE(N > 0 && K > 0 && K <= N && R >= 1 && T >= 1 && T <= PP_TMAX && R * T <= 160 && D % 8 == 0 && ldx % 4 == 0 &&
               (int64_t)N * ldx * 4 < (1ll << 31), "slic_kmeanspp_run_batch: need 1 <= T <= %d, R * T <= 160, D %% 8 == 0, N * ldx * 4 < 2 GiB", PP_TMAX);
  for (int r = 0; r < R; ++r) SLIC_REQUIRE(firsts[r] >= 0 && firsts[r] < N, "slic_kmeanspp_run_batch: firsts[%d] out of range", r);
  hipStream_t st = S(stream);
  const int64_t nblk = slic_cdiv(N, KPP_CH);
  SlicCarver w(workspace);
  float* nd = w.take<float>((size_t)2 * R * T * N);
  double* bpart = w.take<double>((size_t)nblk * R * T);
  double* csum = w.take<double>((size_t)R * nblk);
  double* pot = w.take<double>(R);
  int32_t* candA = w.take<int32_t>((size_t)R * T);
  int32_t* candB = w.take<int32_t>((size_t)R * T);
  int32_t* sel = w.take<int32_t>(R);
  const unsigned nblk_m = (unsigned)slic_cdiv(N, 128);
  auto dist = [&](int RT, int Tc, int Tprev, const int32_t* cand, const float* prev, float* cur) -> int {
    const int ng = (RT + 31) / 32;
    const size_t lds = (size_t)2 * (128 + 32 * ng) * KM_BK * sizeof(float);
#define KPP_BATCH(NG)                                                                                                              \
    {                                                                                                                              \
      static bool attr_set = false;                                                                                                \
      if (!attr_set) {                                                                                                             \
        SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)kpp_dist_mfma_batch<NG>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                           (int)((size_t)2 * (128 + 32 * NG) * KM_BK * sizeof(float))));                           \
        attr_set = true;                                                                                                           \
      }                                                                                                                            \
      kpp_dist_mfma_batch<NG><<<dim3(nblk_m), dim3(256), lds, st>>>(Xp, xnorm, N, D, ldx, cand, RT, Tc, Tprev, prev, sel, cur,     \
                                                                   bpart, nblk);                                                   \
    }
    switch (ng) {
      case 1: KPP_BATCH(1) break;
      case 2: KPP_BATCH(2) break;
      case 3: KPP_BATCH(3) break;
      case 4: KPP_BATCH(4) break;
      default: KPP_BATCH(5) break;
    }
#undef KPP_BATCH
    SLIC_LAUNCH_CHECK();
    return SLIC_OK;
  };
  // centre 0 of every run: one candidate each (the uniformly drawn row), no closest yet
  SLIC_HIP_CHECK(hipMemcpyAsync(candA, firsts, (size_t)R * sizeof(int32_t), hipMemcpyHostToDevice, st));
  int rc = dist(R, 1, 1, candA, nullptr, nd);
  if (rc) return rc;
  const int64_t ustride = (int64_t)(K - 1) * T;
  kpp_select_batch<<<dim3(R), dim3(KPP_ST), 0, st>>>(bpart, nblk, 1, R, T, candA, candB, sel, pot, csum, idx_out, K, nd, N,
                                                     K > 1 ? uniforms : nullptr, ustride);
  SLIC_LAUNCH_CHECK();
  int32_t *cc = candB, *cn2 = candA;
  for (int c = 1; c < K; ++c) {
    const float* prev = nd + (size_t)((c - 1) & 1) * R * T * N;
    float* cur = nd + (size_t)(c & 1) * R * T * N;
    rc = dist(R * T, T, c == 1 ? 1 : T, cc, prev, cur);
    if (rc) return rc;
    kpp_select_batch<<<dim3(R), dim3(KPP_ST), 0, st>>>(bpart, nblk, T, R * T, T, cc, cn2, sel, pot, csum, idx_out + c, K, cur, N,
                                                       c + 1 < K ? uniforms + (size_t)c * T : nullptr, ustride);
    SLIC_LAUNCH_CHECK();
    int32_t* t = cc; cc = cn2; cn2 = t;
  }
  return SLIC_OK;
}

extern "C" size_t slic_cumsum_search_workspace_bytes(int64_t N) {
  return slic_align_up((size_t)slic_cdiv(N, 1024) * 8, 256);
}
extern "C" int slic_cumsum_search(const float* v, int64_t N, const double* vals, int T,
                                  int32_t* idx_out, void* workspace, void* stream) {
  SLIC_REQUIRE(v && vals && idx_out && workspace && N > 0 && T > 0, "slic_cumsum_search: bad args");
  const int64_t nc = slic_cdiv(N, 1024);
  cs_chunk_sums<<<dim3((unsigned)nc), dim3(256), 0, S(stream)>>>(v, N, (double*)workspace);
  SLIC_LAUNCH_CHECK();
  cs_scan_chunks<<<dim3(1), dim3(64), 0, S(stream)>>>((double*)workspace, nc);
  SLIC_LAUNCH_CHECK();
  cs_search<<<dim3((unsigned)slic_cdiv(T, 4)), dim3(256), 0, S(stream)>>>(v, N, (const double*)workspace, nc, vals, T, idx_out);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
