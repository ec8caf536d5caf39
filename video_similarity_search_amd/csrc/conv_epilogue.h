// shared by conv.hip and conv_wino2.hip: the LDS-image epilogue of the convolution kernels
#pragma once
#include "common.h"

// Shared epilogue of the gather-GEMM kernels: bias / affine / addend / mask / ReLU store + deterministic BatchNorm partials.
//
// The accumulators leave the registers through an LDS image of the output tile ([BM][BN] floats, row-major), so that every
// global access of the epilogue is a 16-byte-per-lane buffer op on a contiguous run of a dst row (CPR = BN / 4 lanes cover one
// row: 256-byte runs for BN = 64) — the MFMA accumulator layout itself (a lane owns ONE column, 16 scattered rows) would
// make each of them a 4-byte access in 128-byte pieces, four times the instruction count for the store and for each of the
// addend / mask / bwd_z loads.  Thread t owns chunk cq = t % CPR of rows rr + RPP * pass (RPP = 256 / CPR rows per pass).
// All tensor accesses are range-checked buffer ops with 32-bit byte offsets: a row past M, a chunk past N (N % 4 == 0) or an
// absent optional operand (resource of size 0) is an out-of-range access — loads return 0, stores are dropped.
// Reductions have a fixed order: rows ascending inside a thread, an xor butterfly over the row groups of a wave, then the four
// waves ascending; one slab row per workgroup.
constexpr int conv_epi_lds_floats(int BM, int BN, int NW = 4) { return BM * BN + 2 * NW * BN + BN; }

// Part 2 of the epilogue: everything after the accumulators have been written to the LDS image [BM][BN] (and a barrier passed).
// Also called on its own by kernels that build the image themselves (the Winograd kernel: its image rows are the four outputs
// of each W-tile).  m0 = first GEMM row of the image, a multiple of BM.
// WPAD (the Winograd kernels on a width that is not a multiple of 4): the image rows live in a W-PADDED row space — image row
// m' = (b, t, h) * Wp + w' with Wp = 4 ceil(W / 4) a power of two dividing BM; rows with w' >= W do not exist.  m0 is then the
// padded index of the first row; the GEMM row of image row m' is (m' / Wp) * W + w', and a block holds BM / Wp * W real rows.
// NTHR = threads of the workgroup (256; the 512-thread Winograd workgroup passes 512: more rows per pass, eight wave partials)
// RMAP 2 (the two-dimensional Winograd kernel, conv_wino2.hip): the image holds BM / 8 TILES of 2 rows x 4 columns — image row
// = tile * 8 + hp * 4 + o is output (b, t, 2 h2 + hp, 4 wt + o) of tile (b, t, h2, wt) (tiles counted with wt fastest, ceil(W / 4)
// per row pair, ceil(H / 2) row pairs per frame); outputs past H or W do not exist.  m0 is then the first TILE of the image and
// full_rows the number of real outputs a full block holds (the same for every block: the launch checks it).
template <int BM, int BN, int RMAP = 0, int NTHR = 256>
__device__ __forceinline__ void conv_epilogue_rows(const SlicConvArgs& p, float* lds, int64_t m0, int n0, int tid, int full_rows = 0) {
  constexpr bool WPAD = RMAP == 1, T2D = RMAP == 2;
  const int64_t mblk = T2D ? m0 / (BM / 8) : m0 / BM;   // block index of this workgroup (its slab row in stat_partial / bwd_partial)
  [[maybe_unused]] const int wp_shift = WPAD ? 31 - __builtin_clz((p.Ws + 3) & ~3) : 0;
  [[maybe_unused]] const int64_t bth_all = WPAD ? p.M / p.Ws : 0;
  [[maybe_unused]] const unsigned t2_wq = (unsigned)(p.Ws + 3) >> 2, t2_hq = (unsigned)(p.Hs + 1) >> 1;
  [[maybe_unused]] const int64_t t2_tiles = T2D ? (p.M / ((int64_t)p.Hs * p.Ws)) * t2_hq * t2_wq : 0;
  // image row -> (exists, GEMM row)
  auto row_of = [&](int row, int64_t& m) -> bool {
    if constexpr (T2D) {
      const int64_t tile = m0 + (row >> 3);
      unsigned q = (unsigned)tile;
      const unsigned wt = q % t2_wq; q /= t2_wq;
      const unsigned h2 = q % t2_hq; q /= t2_hq;        // q = frame (b, t)
      const unsigned hr = 2u * h2 + ((unsigned)row >> 2 & 1u), wc = 4u * wt + ((unsigned)row & 3u);
      m = ((int64_t)q * p.Hs + hr) * p.Ws + wc;
      return tile < t2_tiles && hr < (unsigned)p.Hs && wc < (unsigned)p.Ws;
    } else if constexpr (WPAD) {
      const int64_t mp = m0 + row;
      const int64_t bth = mp >> wp_shift;
      const int wq = (int)(mp - (bth << wp_shift));
      m = bth * p.Ws + wq;
      return wq < p.Ws && bth < bth_all;
    } else {
      m = m0 + row;
      return m < p.M;
    }
  };
  constexpr int CPR = BN / 4;       // 16-byte chunks per tile row
  constexpr int RPP = NTHR / CPR;   // rows per pass of the workgroup's threads
  constexpr int NW = NTHR / 64;     // waves
  constexpr int NPASS = BM / RPP;
  static_assert(NTHR % CPR == 0 && BM % RPP == 0, "tile shape");
  float* tile = lds;                       // [BM][BN]: acc + bias
  float* red1 = lds + BM * BN;             // [NW waves][BN]
  float* red2 = red1 + NW * BN;            // [NW waves][BN]
  float* bmean = red2 + NW * BN;           // [BN]
  const int ewave = tid >> 6, elane = tid & 63;
  // sum over the row groups a wave holds for one chunk column (lanes elane, elane ^ CPR, elane ^ 2 CPR, ...): every lane ends up
  // with the same value, added in the same order
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
  const bool want_stats = p.stat_partial != nullptr;
  const bool want_bwd = p.bwd_partial != nullptr;
  // ---- 2. row-major pass: thread = (row group rr, chunk cq)
  constexpr unsigned OOBE = 0xFFFFFF00u;
  const int64_t dst_rows = p.dst_strided ? (p.M / ((int64_t)p.Ga * p.Gb * p.Gc)) * p.Da * p.Db * p.Dc : p.M;
  const unsigned dst_bytes = (unsigned)(((dst_rows - 1) * (int64_t)p.ldo + p.N) * 4);
  const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc((void*)p.dst, 0, dst_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc((void*)p.addend, 0, p.addend ? dst_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_msk = __builtin_amdgcn_make_buffer_rsrc((void*)p.mask_src, 0, p.mask_src ? dst_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bz = __builtin_amdgcn_make_buffer_rsrc((void*)p.bwd_z, 0, p.bwd_z ? dst_bytes : 0, 0x00020000);
  const bool has_mask = p.mask_src != nullptr, do_relu = p.relu != 0;
  const int cq = tid % CPR, rr = tid / CPR;
  const int n = n0 + cq * 4;
  const bool nv = n < p.N;                       // N % 4 == 0: a chunk is inside or outside as a whole
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f}, bmu = sh, bis = sh;
  if (nv) {
    if (p.scale) sc = *(const f32x4*)(p.scale + n);
    if (p.shift) sh = *(const f32x4*)(p.shift + n);
    if (want_bwd) { bmu = *(const f32x4*)(p.bwd_mean + n); bis = *(const f32x4*)(p.bwd_invstd + n); }
  }
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1, fs = s1;
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int row = ps * RPP + rr;
    int64_t m;
    const bool ok = row_of(row, m) && nv;
    unsigned ro;
    if (p.dst_strided) {
      unsigned q = (unsigned)m;
      const unsigned gc = q % (unsigned)p.Gc; q /= (unsigned)p.Gc;
      const unsigned gbb = q % (unsigned)p.Gb; q /= (unsigned)p.Gb;
      const unsigned gaa = q % (unsigned)p.Ga; q /= (unsigned)p.Ga;
      ro = ((((q * p.Da + gaa * p.da + p.ea) * p.Db + gbb * p.db + p.eb) * p.Dc + gc * p.dc + p.ec) * (unsigned)p.ldo) * 4u;
    } else {
      ro = (unsigned)m * (unsigned)(p.ldo * 4);
    }
    const unsigned off = ok ? ro + (unsigned)n * 4u : OOBE;
    f32x4 v = *(const f32x4*)&tile[row * BN + cq * 4];
    if (ok) fs += v;
    v = v * sc + sh;
    v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_add, off, 0, 0));
    const f32x4 mk = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_msk, off, 0, 0));
    const f32x4 zz = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bz, off, 0, 0));
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float x = v[c];
      x = (has_mask && !(mk[c] > 0.f)) ? 0.f : x;
      x = do_relu ? fmaxf(x, 0.f) : x;
      v[c] = x;
    }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rs_dst, off, 0, 0);
    if (ok) {
      s1 += v;
      s2 += v * ((zz - bmu) * bis);
    }
  }
  if (want_bwd) {
    // BatchNorm-backward partials of the stored gradient: (sum v, sum v * xhat) per channel over this row block
    s1 = wave_rows_sum(s1);
    s2 = wave_rows_sum(s2);
    if (elane < CPR) {
      *(f32x4*)&red1[ewave * BN + cq * 4] = s1;
      *(f32x4*)&red2[ewave * BN + cq * 4] = s2;
    }
    __syncthreads();
    if (tid < BN) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { t1 += red1[w * BN + tid]; t2 += red2[w * BN + tid]; }
      const int nn = n0 + tid;
      if (nn < p.N) {
        p.bwd_partial[(mblk * 2 + 0) * p.N + nn] = t1;
        p.bwd_partial[(mblk * 2 + 1) * p.N + nn] = t2;
      }
    }
  }
  if (want_stats) {
    // BatchNorm partials of v = acc + bias over this workgroup's valid rows, per channel:
    //   slab[blk][0][n] = sum v          slab[blk][1][n] = sum (v - mean_blk)^2   (second pass over the LDS image,
    // so the variance never comes from E[x^2] - mean^2); bn_finalize merges workgroups with Chan's formula in double.
    int64_t left = p.M - m0, full = BM;
    if constexpr (WPAD) {
      full = (BM >> wp_shift) * p.Ws;                 // real rows of a full block
      left = p.M - mblk * full;
    }
    if constexpr (T2D) {
      full = full_rows;
      left = p.M - mblk * full;
    }
    const float inv_rows = 1.0f / (float)(left < full ? left : full);
    if (want_bwd) __syncthreads();             // red1 is still being read by the block above
    fs = wave_rows_sum(fs);
    if (elane < CPR) *(f32x4*)&red1[ewave * BN + cq * 4] = fs;
    __syncthreads();
    if (tid < BN) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red1[w * BN + tid];
      bmean[tid] = t * inv_rows;
      const int nn = n0 + tid;
      if (nn < p.N) p.stat_partial[(mblk * 2 + 0) * p.N + nn] = t;
    }
    __syncthreads();
    const f32x4 mu = *(const f32x4*)&bmean[cq * 4];
    f32x4 q2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int row = ps * RPP + rr;
      int64_t mm;
      if (row_of(row, mm)) {
        const f32x4 d = *(const f32x4*)&tile[row * BN + cq * 4] - mu;
        q2 += d * d;
      }
    }
    q2 = wave_rows_sum(q2);
    if (elane < CPR) *(f32x4*)&red2[ewave * BN + cq * 4] = q2;
    __syncthreads();
    if (tid < BN) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red2[w * BN + tid];
      const int nn = n0 + tid;
      if (nn < p.N) p.stat_partial[(mblk * 2 + 1) * p.N + nn] = t;
    }
  }
}

