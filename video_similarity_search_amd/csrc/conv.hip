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
// Epilogue (fused): + bias, per-channel affine (eval-mode BN), ReLU, + addend (residual / gradient
// accumulation), and per-workgroup BatchNorm partial sums (sum, sum of squares per channel) written
// to a slab that bn.hip reduces in fixed order (deterministic train-mode statistics).
#include "common.h"
#include "conv_common.h"

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const SlicConvArgs p) {
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

  // ---- staging setup: thread owns chunk column cq of rows srow + 32 i
  const int cq = tid & 7, srow = tid >> 3;
  int64_t abase[AL];
  int a0[AL], b0[AL], c0[AL];
#pragma unroll
  for (int i = 0; i < AL; ++i) {
    const int64_t m = m0 + srow + 32 * i;
    if (m < p.M) {
      int64_t r = m;
      const int gc = (int)(r % p.Gc); r /= p.Gc;
      const int gb = (int)(r % p.Gb); r /= p.Gb;
      const int ga = (int)(r % p.Ga); r /= p.Ga;   // r = batch
      a0[i] = ga * p.sa; b0[i] = gb * p.sb; c0[i] = gc * p.sc;
      abase[i] = ((((int64_t)r * p.Ts + a0[i]) * p.Hs + b0[i]) * p.Ws + c0[i]) * p.Cs;
    } else {
      a0[i] = -(1 << 20); b0[i] = 0; c0[i] = 0; abase[i] = 0;
    }
  }
  const float* wrow[BL];
  bool wvalid[BL];
#pragma unroll
  for (int i = 0; i < BL; ++i) {
    const int n = n0 + srow + 32 * i;
    wvalid[i] = n < p.N;
    wrow[i] = p.wgt + (int64_t)(wvalid[i] ? n : 0) * p.ldw;
  }
  f32x4 ga[AL], gb[BL];
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  auto gload = [&](int kt) {
    const int4 e = ((const int4*)p.tab)[kt * 8 + cq];
    const bool cv = e.y >= 0;
    const int oa = (e.y & 255) - 128, ob = ((e.y >> 8) & 255) - 128, oc = ((e.y >> 16) & 255) - 128;
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const bool ok = cv && (unsigned)(a0[i] + oa) < (unsigned)p.Ts &&
                      (unsigned)(b0[i] + ob) < (unsigned)p.Hs && (unsigned)(c0[i] + oc) < (unsigned)p.Ws;
      ga[i] = ok ? *(const f32x4*)(p.src + abase[i] + e.x) : z4;
    }
#pragma unroll
    for (int i = 0; i < BL; ++i) gb[i] = (cv && wvalid[i]) ? *(const f32x4*)(wrow[i] + e.z) : z4;
  };
  auto lwrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AL; ++i) *(f32x4*)&As[buf * BM * 32 + cv_off(srow + 32 * i, cq)] = ga[i];
#pragma unroll
    for (int i = 0; i < BL; ++i) *(f32x4*)&Bs[buf * BN * 32 + cv_off(srow + 32 * i, cq)] = gb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

  const int r = lane & 31, h = lane >> 5;
  const int nk = p.nchunks >> 3;
  if (nk > 0) {
    gload(0);
    lwrite(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const float* Ab = As + buf * BM * 32;
    const float* Bb = Bs + buf * BN * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *(const f32x4*)&Ab[cv_off(wm * WTM + i * 32 + r, 2 * q + h)];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *(const f32x4*)&Bb[cv_off(wn * WTN + j * 32 + r, 2 * q + h)];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) lwrite(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue
  const bool want_stats = p.stat_partial != nullptr;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WTN + j * 32 + r;
    const bool nv = n < p.N;
    const float bias = (p.bias && nv) ? p.bias[n] : 0.f;
    const float sc = (p.scale && nv) ? p.scale[n] : 1.f;
    const float sh = (p.shift && nv) ? p.shift[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int64_t m = m0 + wm * WTM + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
        if (m < p.M && nv) {
          int64_t off;
          if (p.dst_strided) {
            int64_t q = m;
            const int gc = (int)(q % p.Gc); q /= p.Gc;
            const int gbb = (int)(q % p.Gb); q /= p.Gb;
            const int gaa = (int)(q % p.Ga); q /= p.Ga;
            off = ((((int64_t)q * p.Da + gaa * p.da + p.ea) * p.Db + gbb * p.db + p.eb) * p.Dc +
                   gc * p.dc + p.ec) * (int64_t)p.ldo + n;
          } else {
            off = m * (int64_t)p.ldo + n;
          }
          float v = (acc[i][j][g] + bias) * sc + sh;
          if (p.addend) v += p.addend[off];
          if (p.relu) v = fmaxf(v, 0.f);
          p.dst[off] = v;
        }
      }
    }
  }
  if (want_stats) {
    // BatchNorm partials of v = acc + bias over this workgroup's valid rows, per channel:
    //   slab[blk][0][n] = sum v          slab[blk][1][n] = sum (v - mean_blk)^2   (two passes over registers,
    // so the variance never comes from E[x^2] - mean^2); bn_finalize merges workgroups with Chan's formula in double.
    // Reduction order is fixed: lane halves -> waves along M -> one slab row per workgroup.
    __syncthreads();
    float* red = lds;             // [WM][BN]
    float* bmean = lds + WM * BN; // [BN]
    const int64_t left = p.M - m0;
    const float inv_rows = 1.0f / (float)(left < BM ? left : BM);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = wn * WTN + j * 32 + r;
        const int n = n0 + col;
        const float bias = (p.bias && n < p.N) ? p.bias[n] : 0.f;
        const float mu = pass ? bmean[col] : 0.f;
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            const int64_t m = m0 + wm * WTM + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
            if (m < p.M) {
              const float v = acc[i][j][g] + bias;
              a += pass ? (v - mu) * (v - mu) : v;
            }
          }
        a += __shfl_xor(a, 32);
        if (h == 0) red[wm * BN + col] = a;
      }
      __syncthreads();
      if (tid < BN) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) t += red[w * BN + tid];
        if (!pass) bmean[tid] = t * inv_rows;
        const int n = n0 + tid;
        if (n < p.N) p.stat_partial[((int64_t)blockIdx.x * 2 + pass) * p.N + n] = t;
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------
// Weight gradient: dWp[n][kidx] = sum_m A[m][kidx] * dY[m][n]   (A = the forward's gathered rows).
// Workgroup = 64 kidx x 64 n, waves 2 x 2 (32 x 32 each), reduction over a slice of m in chunks of
// 32 positions.  Both LDS tiles are [32 m][64] (channel-contiguous, as they sit in HBM); the MFMA
// operands are read with ds_read_b32 (lanes = 32 consecutive channels: conflict-free) — one read
// per 64-cycle MFMA, far below the LDS rate.  Split over m: gridDim.z slabs, reduced (and unpacked
// into the reference [N][Cin][taps] layout) by conv_wgrad_reduce in fixed order.
// ------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const SlicConvArgs p, const float* __restrict__ dy,
                                                         int ldy, float* __restrict__ slab,
                                                         int m_per_split) {
  __shared__ __attribute__((aligned(16))) float Xs[2][G][32 * 64];
  __shared__ __attribute__((aligned(16))) float Ys[2][32 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave >> 1, wn = wave & 1;
  const int kc0 = blockIdx.x * (16 * G);   // first 16-byte chunk of this workgroup's kidx range
  const int n0 = blockIdx.y * 64;
  const int64_t mbeg = (int64_t)blockIdx.z * m_per_split;
  int64_t mend = mbeg + m_per_split;
  if (mend > p.M) mend = p.M;

  // staging: thread owns chunk column cq (16 chunks = 64 floats per row), rows srow, srow + 16
  const int cq = tid & 15, srow = tid >> 4;
  int4 e[G];
  bool cv[G];
  int oa[G], ob[G], oc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int q = kc0 + g * 16 + cq;
    e[g] = q < p.nchunks ? ((const int4*)p.tab)[q] : make_int4(0, -1, 0, 0);
    cv[g] = e[g].y >= 0;
    oa[g] = (e[g].y & 255) - 128; ob[g] = ((e[g].y >> 8) & 255) - 128; oc[g] = ((e[g].y >> 16) & 255) - 128;
  }
  const bool nvalid = (n0 + cq * 4) < p.N;
  f32x4 gx[G][2], gy[2];
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  auto gload = [&](int64_t mt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t m = mt + srow + 16 * i;
      if (m < mend) {
        int64_t rr = m;
        const int gc = (int)(rr % p.Gc); rr /= p.Gc;
        const int gbb = (int)(rr % p.Gb); rr /= p.Gb;
        const int gaa = (int)(rr % p.Ga); rr /= p.Ga;
        const int a0 = gaa * p.sa, b0 = gbb * p.sb, c0 = gc * p.sc;
        const int64_t base = ((((int64_t)rr * p.Ts + a0) * p.Hs + b0) * p.Ws + c0) * p.Cs;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const bool ok = cv[g] && (unsigned)(a0 + oa[g]) < (unsigned)p.Ts &&
                          (unsigned)(b0 + ob[g]) < (unsigned)p.Hs && (unsigned)(c0 + oc[g]) < (unsigned)p.Ws;
          gx[g][i] = ok ? *(const f32x4*)(p.src + base + e[g].x) : z4;
        }
        gy[i] = nvalid ? *(const f32x4*)(dy + m * (int64_t)ldy + n0 + cq * 4) : z4;
      } else {
#pragma unroll
        for (int g = 0; g < G; ++g) gx[g][i] = z4;
        gy[i] = z4;
      }
    }
  };
  auto lwrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = srow + 16 * i;
#pragma unroll
      for (int g = 0; g < G; ++g) *(f32x4*)&Xs[buf][g][row * 64 + cq * 4] = gx[g][i];
      *(f32x4*)&Ys[buf][row * 64 + cq * 4] = gy[i];
    }
  };
  f32x16 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[g][v] = 0.f;
  const int r = lane & 31, h = lane >> 5;
  const int64_t nmt = (mend > mbeg) ? (mend - mbeg + 31) / 32 : 0;
  if (nmt > 0) {
    gload(mbeg);
    lwrite(0);
  }
  __syncthreads();
  for (int64_t t = 0; t < nmt; ++t) {
    const int buf = (int)(t & 1);
    if (t + 1 < nmt) gload(mbeg + (t + 1) * 32);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float b = Ys[buf][(2 * s + h) * 64 + wn * 32 + r];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float a = Xs[buf][g][(2 * s + h) * 64 + wk * 32 + r];
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g], 0, 0, 0);
      }
    }
    if (t + 1 < nmt) lwrite(buf ^ 1);
    __syncthreads();
  }
  // slab[z][n][kidx]: rows of the accumulator = kidx, cols = n
  const int Kp = p.nchunks * 4;
  float* out = slab + (int64_t)blockIdx.z * p.N * Kp;
  const int n = n0 + wn * 32 + r;
  if (n < p.N) {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int kidx = (kc0 + g * 16) * 4 + wk * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
        if (kidx < Kp) out[(int64_t)n * Kp + kidx] = acc[g][v];
      }
  }
}

// dW[n][c][tap] (reference layout, C = real channel count) = sum over splits of slab[s][n][tap*Cs + c]
__global__ void conv_wgrad_reduce(const float* __restrict__ slab, int S, int N, int Kp, int Cs, int C,
                                  int ntaps, float* __restrict__ dW) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t tot = (int64_t)N * C * ntaps;
  if (e >= tot) return;
  const int tap = (int)(e % ntaps);
  const int c = (int)((e / ntaps) % C);
  const int n = (int)(e / ((int64_t)ntaps * C));
  const int64_t src = (int64_t)n * Kp + tap * Cs + c;
  float a = 0.f;
  for (int s = 0; s < S; ++s) a += slab[(int64_t)s * N * Kp + src];
  dW[e] = a;
}

// ---- weight packing -------------------------------------------------------------------------
// forward:  Wp[n][tap*Cs + c] = W[n][c][tap]  (zero for c >= C and the K padding)
__global__ void pack_w_fwd(const float* __restrict__ W, int N, int C, int ntaps, int Cs, int Kp,
                           float* __restrict__ Wp) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)N * Kp) return;
  const int k = (int)(e % Kp), n = (int)(e / Kp);
  const int tap = k / Cs, c = k % Cs;
  Wp[e] = (tap < ntaps && c < C) ? W[((int64_t)n * C + c) * ntaps + tap] : 0.f;
}
// data gradient:  Wd[c][tap*N + n] = W[n][c][tap]   (rows = input channels, Cs rows, zero padded)
__global__ void pack_w_dgrad(const float* __restrict__ W, int N, int C, int ntaps, int Cs, int Kd,
                             float* __restrict__ Wd) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)Cs * Kd) return;
  const int k = (int)(e % Kd), c = (int)(e / Kd);
  const int tap = k / N, n = k % N;
  Wd[e] = (tap < ntaps && c < C) ? W[((int64_t)n * C + c) * ntaps + tap] : 0.f;
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

// ------------------------------------ C ABI ------------------------------------------------
static inline hipStream_t S_(void* s) { return (hipStream_t)s; }

static int validate(const SlicConvArgs* a, const char* who) {
  SLIC_REQUIRE(a && a->src && a->tab, "%s: null pointer", who);
  SLIC_REQUIRE(a->Cs > 0 && a->Cs % 4 == 0, "%s: source channels must be a multiple of 4 (Cs=%d)", who, a->Cs);
  SLIC_REQUIRE(a->nchunks >= 0 && a->nchunks % 8 == 0, "%s: nchunks %% 8 != 0 (%d)", who, a->nchunks);
  SLIC_REQUIRE(a->M > 0 && a->N > 0 && a->Ga > 0 && a->Gb > 0 && a->Gc > 0 && a->Ts > 0 && a->Hs > 0 && a->Ws > 0,
               "%s: bad shape", who);
  SLIC_REQUIRE(a->M % ((int64_t)a->Ga * a->Gb * a->Gc) == 0, "%s: M is not batch * grid", who);
  SLIC_REQUIRE(((uintptr_t)a->src % 16) == 0, "%s: src not 16-byte aligned", who);
  return SLIC_OK;
}

template <int BM, int BN, int WM, int WN>
static int launch_gemm(const SlicConvArgs& a, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * (BM + BN) * 32 * sizeof(float);
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

extern "C" int slic_conv_tile_m(const SlicConvArgs* a, int variant) {
  // rows per workgroup for the tile the dispatcher picks (callers size stat_partial with it)
  if (variant == 1) return 128;
  if (variant == 2) return 64;
  if (variant == 3) return 256;
  if (!a) return 128;
  const int64_t blocks128 = slic_cdiv(a->M, 128) * slic_cdiv(a->N, a->N > 64 ? 128 : 64);
  return blocks128 >= 512 ? 128 : 64;
}

extern "C" int slic_conv_gemm(const SlicConvArgs* a, int variant, void* stream) {
  int rc = validate(a, "slic_conv_gemm");
  if (rc) return rc;
  SLIC_REQUIRE(a->wgt && a->dst && a->ldw % 4 == 0 && a->ldo >= 1, "slic_conv_gemm: bad weight/dst");
  SLIC_REQUIRE(((uintptr_t)a->wgt % 16) == 0, "slic_conv_gemm: wgt not 16-byte aligned");
  hipStream_t st = S_(stream);
  const int bm = slic_conv_tile_m(a, variant);
  if (bm == 256) return launch_gemm<256, 64, 4, 1>(*a, st);
  if (bm == 128) {
    if (a->N > 64) return launch_gemm<128, 128, 2, 2>(*a, st);
    return launch_gemm<128, 64, 2, 2>(*a, st);
  }
  return launch_gemm<64, 64, 2, 2>(*a, st);
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
  SLIC_REQUIRE(ntaps * a->Cs <= a->nchunks * 4, "slic_conv_wgrad: table shorter than ntaps*Cs");
  hipStream_t st = S_(stream);
  float* slab = (float*)workspace;
  int64_t per = slic_cdiv(a->M, splits);
  per = slic_cdiv(per, 32) * 32;
  const int S = (int)slic_cdiv(a->M, per);
  const int Kp = a->nchunks * 4;
  constexpr int G = 2;
  dim3 grid((unsigned)slic_cdiv(a->nchunks, 16 * G), (unsigned)slic_cdiv(a->N, 64), (unsigned)S);
  conv_wgrad_kernel<G><<<grid, dim3(256), 0, st>>>(*a, dy, ldy, slab, (int)per);
  SLIC_LAUNCH_CHECK();
  const int64_t tot = (int64_t)a->N * C * ntaps;
  conv_wgrad_reduce<<<dim3((unsigned)slic_cdiv(tot, 256)), dim3(256), 0, st>>>(slab, S, a->N, Kp, a->Cs, C, ntaps, dW);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_pack_weight_fwd(const float* W, int N, int C, int ntaps, int Cs, int Kp, float* Wp,
                                    void* stream) {
  SLIC_REQUIRE(W && Wp && N > 0 && C > 0 && ntaps > 0 && Cs >= C && Kp >= ntaps * Cs, "slic_pack_weight_fwd: bad args");
  const int64_t tot = (int64_t)N * Kp;
  pack_w_fwd<<<dim3((unsigned)slic_cdiv(tot, 256)), dim3(256), 0, S_(stream)>>>(W, N, C, ntaps, Cs, Kp, Wp);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_pack_weight_dgrad(const float* W, int N, int C, int ntaps, int Cs, int Kd, float* Wd,
                                      void* stream) {
  SLIC_REQUIRE(W && Wd && N > 0 && C > 0 && ntaps > 0 && Cs >= C && Kd >= ntaps * N, "slic_pack_weight_dgrad: bad args");
  const int64_t tot = (int64_t)Cs * Kd;
  pack_w_dgrad<<<dim3((unsigned)slic_cdiv(tot, 256)), dim3(256), 0, S_(stream)>>>(W, N, C, ntaps, Cs, Kd, Wd);
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
