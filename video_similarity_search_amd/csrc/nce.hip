// Memory-bank NCE (CMC-style) — the HBM-bound gather/dot/softmax path.
//
// Replaces NCEAverage.forward + NCESoftmaxLoss of /root/reference/loss/NCE_loss.py:26-88, 341-352
// (called from contrastive_train_epoch, online_train.py:175-190):
//   index_select(bank, idx) [B, K+1, D] + bmm + /T     -> nce_scores_fwd: rows are gathered straight into
//        registers (32 lanes x float4 per 128-float row, 2 rows per wave step) and reduced with shuffles;
//        the 16.8 MB gathered copy per bank at B = 32, K = 1024 is never materialised.
//   autograd of the bmm wrt the features                 -> nce_scores_bwd (bank rows are detached, :41,46)
//   momentum update + renormalise + index_copy_          -> nce_bank_update
//   CrossEntropyLoss(x.squeeze(), zeros)                 -> softmax_ce0_fwd / _bwd (row log-sum-exp, class 0)
// Algorithmic bytes: B*(K+1)*D*4 gathered per bank per forward (and again per backward).
#include "common.h"
#include <math.h>

// out[b, j] = <bank[idx[b, j]], f[b]> / T.   grid: (ceil((K1)/rows_per_block), B)
__global__ __launch_bounds__(256) void nce_scores_fwd(const float* __restrict__ bank, const int64_t* __restrict__ idx,
                                                      const float* __restrict__ f, int K1, int D, float invT,
                                                      float* __restrict__ out, float* __restrict__ gathered) {
  const int b = blockIdx.y;
  const int sub = threadIdx.x >> 5, l = threadIdx.x & 31;     // 8 row-slots of 32 lanes
  const float* fb = f + (int64_t)b * D;
  for (int j = blockIdx.x * 8 + sub; j < K1; j += gridDim.x * 8) {
    const float* row = bank + idx[(int64_t)b * K1 + j] * (int64_t)D;
    float a = 0.f;
    for (int k = l * 4; k < D; k += 128) {
      const f32x4 r = *(const f32x4*)(row + k);
      const f32x4 x = *(const f32x4*)(fb + k);
      if (gathered) *(f32x4*)(gathered + ((int64_t)b * K1 + j) * D + k) = r;   // rows as scored, for the backward
      a = fmaf(r.x, x.x, a); a = fmaf(r.y, x.y, a); a = fmaf(r.z, x.z, a); a = fmaf(r.w, x.w, a);
    }
    for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (l == 0) out[(int64_t)b * K1 + j] = a * invT;
  }
}

// df[b, :] = (1/T) sum_j dout[b, j] * bank[idx[b, j], :]     one workgroup per b, j ascending per thread column
__global__ __launch_bounds__(256) void nce_scores_bwd(const float* __restrict__ bank, const int64_t* __restrict__ idx,
                                                      const float* __restrict__ dout, int K1, int D, float invT,
                                                      float* __restrict__ df) {
  __shared__ float red[8][512];
  const int b = blockIdx.x;
  const int sub = threadIdx.x >> 5, l = threadIdx.x & 31;
  for (int k0 = 0; k0 < D; k0 += 128) {
    const int k = k0 + l * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (k < D) {
      int j = sub;
      for (; j + 24 < K1; j += 32) {                       // four gathers in flight; the adds stay j-ascending
        const int64_t e = (int64_t)b * K1 + j;
        const float g0 = dout[e], g1 = dout[e + 8], g2 = dout[e + 16], g3 = dout[e + 24];
        const f32x4 r0 = *(const f32x4*)(bank + idx[e] * (int64_t)D + k);
        const f32x4 r1 = *(const f32x4*)(bank + idx[e + 8] * (int64_t)D + k);
        const f32x4 r2 = *(const f32x4*)(bank + idx[e + 16] * (int64_t)D + k);
        const f32x4 r3 = *(const f32x4*)(bank + idx[e + 24] * (int64_t)D + k);
        a += r0 * g0; a += r1 * g1; a += r2 * g2; a += r3 * g3;
      }
      for (; j < K1; j += 8) {
        const float g = dout[(int64_t)b * K1 + j];
        const f32x4 r = *(const f32x4*)(bank + idx[(int64_t)b * K1 + j] * (int64_t)D + k);
        a += r * g;
      }
    }
    *(f32x4*)&red[sub][l * 4] = a;
    __syncthreads();
    if (threadIdx.x < 128 && k0 + threadIdx.x < D) {
      float t = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) t += red[s][threadIdx.x];
      df[(int64_t)b * D + k0 + threadIdx.x] = t * invT;
    }
    __syncthreads();
  }
}

// bank[y[b]] <- normalise(m * bank[y[b]] + (1 - m) * f[b])   one wave per b (index_copy_: later b wins on duplicates)
__global__ void nce_bank_update(float* __restrict__ bank, const int64_t* __restrict__ y, const float* __restrict__ f,
                                int B, int D, float momentum) {
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  const int lane = threadIdx.x & 63;
  float* row = bank + y[b] * (int64_t)D;
  const float* fb = f + (int64_t)b * D;
  float s = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float v = row[k] * momentum + fb[k] * (1.f - momentum);
    s = fmaf(v, v, s);
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float nrm = sqrtf(s);
  for (int k = lane; k < D; k += 64) row[k] = (row[k] * momentum + fb[k] * (1.f - momentum)) / nrm;
}

// rowloss[b] = logsumexp(x[b, :]) - x[b, 0];  lse kept for the backward
__global__ __launch_bounds__(256) void softmax_ce0_fwd(const float* __restrict__ x, int K1, float* __restrict__ lse,
                                                       float* __restrict__ rowloss) {
  __shared__ float red[4];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* xb = x + (int64_t)b * K1;
  float m = -INFINITY;
  for (int j = t; j < K1; j += 256) m = fmaxf(m, xb[j]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((t & 63) == 0) red[t >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int j = t; j < K1; j += 256) s += expf(xb[j] - m);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  if (t == 0) {
    const float l = m + logf(red[0] + red[1] + red[2] + red[3]);
    lse[b] = l;
    rowloss[b] = l - xb[0];
  }
}
__global__ void softmax_ce0_bwd(const float* __restrict__ x, const float* __restrict__ lse, int B, int K1,
                                const float* __restrict__ gscale, float* __restrict__ dx) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)B * K1) return;
  const int b = (int)(e / K1), j = (int)(e % K1);
  const float g = (gscale ? *gscale : 1.f) / (float)B;
  dx[e] = (expf(x[e] - lse[b]) - (j == 0 ? 1.f : 0.f)) * g;
}
__global__ void mean_rows_serial(const float* __restrict__ v, int n, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double a = 0.0;
    for (int i = 0; i < n; ++i) a += (double)v[i];
    *out = (float)(a / (double)n);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The whole contrastive step of online_train.py:175-190 — contrast(feat_l, feat_ab, index) + NCESoftmaxLoss on both outputs —
// in THREE launches (the path is launch-bound: 33.6 MB of gathered rows take microseconds, a dozen launches 0.4 ms):
//   nce_fused_fwd   (b, side): gather + dot + /T for the side's bank, the rows kept as scored, and the row's softmax cross-entropy
//                   against class 0 (online max / sum-exp per 32-lane group, the eight groups merged in a fixed order)
//   nce_fused_update: momentum update + renormalise of both banks, and loss = mean_b rowloss[0] + mean_b rowloss[1] (serial, double)
//   nce_fused_bwd   (b, side): dfeat[b] = gscale / (B T) sum_j (exp(s_j - lse_b) - [j = 0]) rows[b, j]   (j ascending per column)
// side 0: feature ab scored against memory_l (-> out_ab), side 1: feature l against memory_ab (-> out_l)  (NCE_loss.py:41-48).
// ---------------------------------------------------------------------------------------------------------------
#define NCE_PARTS 16                 // a score row's K1 entries are split over this many workgroups (slic_hip.h: SLIC_NCE_PARTS)
__global__ __launch_bounds__(256) void nce_fused_fwd(const float* __restrict__ bank0, const float* __restrict__ bank1,
                                                     const float* __restrict__ f0, const float* __restrict__ f1,
                                                     const int64_t* __restrict__ idx, int K1, int D, float invT,
                                                     float* __restrict__ scores /* [2][B][K1] */, float* __restrict__ rows /* [2][B][K1][D] */,
                                                     float* __restrict__ part /* [2][B][NCE_PARTS][2] */) {
  __shared__ float rm[8], rs[8];
  const int b = blockIdx.x, side = blockIdx.y, c = blockIdx.z, B = gridDim.x;
  const float* bank = side ? bank1 : bank0;
  const float* fb = (side ? f1 : f0) + (int64_t)b * D;
  const int sub = threadIdx.x >> 5, l = threadIdx.x & 31;
  float* sc = scores + ((int64_t)side * B + b) * K1;
  float* rw = rows + ((int64_t)side * B + b) * K1 * (int64_t)D;
  const int64_t* ix = idx + (int64_t)b * K1;
  const int cl = (K1 + NCE_PARTS - 1) / NCE_PARTS, j1 = min(K1, (c + 1) * cl);
  float m = -INFINITY, sum = 0.f;
  for (int j = c * cl + sub; j < j1; j += 8) {
    const float* row = bank + ix[j] * (int64_t)D;
    float a = 0.f;
    for (int k = l * 4; k < D; k += 128) {
      const f32x4 r = *(const f32x4*)(row + k);
      const f32x4 x = *(const f32x4*)(fb + k);
      *(f32x4*)(rw + (int64_t)j * D + k) = r;
      a = fmaf(r.x, x.x, a); a = fmaf(r.y, x.y, a); a = fmaf(r.z, x.z, a); a = fmaf(r.w, x.w, a);
    }
    for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o);
    const float v = a * invT;
    if (l == 0) sc[j] = v;
    const float mn = fmaxf(m, v);                             // online log-sum-exp (every lane of the group holds the same values)
    sum = sum * expf(m - mn) + expf(v - mn);
    m = mn;
  }
  if (l == 0) { rm[sub] = m; rs[sub] = sum; }
  __syncthreads();
  if (threadIdx.x == 0) {                                      // the eight groups in a fixed order; an idle group has m = -inf, sum = 0
    float M = rm[0];
    for (int g = 1; g < 8; ++g) M = fmaxf(M, rm[g]);
    float S = 0.f;
    if (M > -INFINITY)
      for (int g = 0; g < 8; ++g) S += rs[g] * expf(rm[g] - M);
    float* pp = part + (((int64_t)side * B + b) * NCE_PARTS + c) * 2;
    pp[0] = M; pp[1] = S;
  }
}

__global__ void nce_fused_update(float* __restrict__ bank_l, float* __restrict__ bank_ab, const int64_t* __restrict__ y,
                                 const float* __restrict__ fl, const float* __restrict__ fab, int B, int D, float momentum,
                                 const float* __restrict__ part, const float* __restrict__ scores, int K1,
                                 float* __restrict__ lse, float* __restrict__ rowloss, float* __restrict__ loss) {
  const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);     // wave = (bank, b)
  const int lane = threadIdx.x & 63;
  if (w < 2 * B) {
    const int which = w / B, b = w - which * B;
    float* row = (which ? bank_ab : bank_l) + y[b] * (int64_t)D;
    const float* fb = (which ? fab : fl) + (int64_t)b * D;
    // index_copy_ semantics on duplicate labels: the LAST b wins, and every b reads the bank as it was before the call
    bool last = true;
    for (int b2 = b + 1; b2 < B; ++b2) last = last && (y[b2] != y[b]);
    float sq = 0.f;
    for (int k = lane; k < D; k += 64) {
      const float v = row[k] * momentum + fb[k] * (1.f - momentum);
      sq = fmaf(v, v, sq);
    }
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float nrm = sqrtf(sq);
    if (last)
      for (int k = lane; k < D; k += 64) row[k] = (row[k] * momentum + fb[k] * (1.f - momentum)) / nrm;
  }
  if (blockIdx.x == 0 && part) {
    // a score row's log-sum-exp from its NCE_PARTS pieces (ascending), its cross-entropy against class 0, then the two means
    for (int r = threadIdx.x; r < 2 * B; r += blockDim.x) {
      const float* pp = part + (int64_t)r * NCE_PARTS * 2;
      float M = pp[0];
      for (int c = 1; c < NCE_PARTS; ++c) M = fmaxf(M, pp[2 * c]);
      float S = 0.f;
      for (int c = 0; c < NCE_PARTS; ++c)
        if (pp[2 * c] > -INFINITY) S += pp[2 * c + 1] * expf(pp[2 * c] - M);
      const float L = M + logf(S);
      lse[r] = L;
      rowloss[r] = L - scores[(int64_t)r * K1];
    }
    __syncthreads();
    if (threadIdx.x == 0 && loss) {
      double a0 = 0.0, a1 = 0.0;
      for (int i = 0; i < B; ++i) { a0 += (double)rowloss[i]; a1 += (double)rowloss[B + i]; }
      *loss = (float)(a0 / (double)B) + (float)(a1 / (double)B);
    }
  }
}

// (b, side, 128-column slice): 32 groups of 32 lanes stride the K1 rows, four rows in flight per group; a group adds j-ascending and
// the groups are summed in ascending order
__global__ __launch_bounds__(1024) void nce_fused_bwd(const float* __restrict__ rows, const float* __restrict__ scores,
                                                      const float* __restrict__ lse, int K1, int D, float invT,
                                                      const float* __restrict__ gscale, float* __restrict__ df /* [2][B][D] */) {
  __shared__ float red[32][128];
  const int b = blockIdx.x, side = blockIdx.y, B = gridDim.x;
  const int sub = threadIdx.x >> 5, l = threadIdx.x & 31;
  const float* sc = scores + ((int64_t)side * B + b) * K1;
  const float* rw = rows + ((int64_t)side * B + b) * K1 * (int64_t)D;
  const float L = lse[side * B + b];
  const float g = (gscale ? *gscale : 1.f) / (float)B;
  const int k0 = blockIdx.z * 128, k = k0 + l * 4;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (k < D) {
    int j = sub;
    for (; j + 96 < K1; j += 128) {
      const float p0 = (expf(sc[j] - L) - (j == 0 ? 1.f : 0.f)) * g, p1 = expf(sc[j + 32] - L) * g;
      const float p2 = expf(sc[j + 64] - L) * g, p3 = expf(sc[j + 96] - L) * g;
      const f32x4 r0 = *(const f32x4*)(rw + (int64_t)j * D + k), r1 = *(const f32x4*)(rw + (int64_t)(j + 32) * D + k);
      const f32x4 r2 = *(const f32x4*)(rw + (int64_t)(j + 64) * D + k), r3 = *(const f32x4*)(rw + (int64_t)(j + 96) * D + k);
      a += r0 * p0; a += r1 * p1; a += r2 * p2; a += r3 * p3;
    }
    for (; j < K1; j += 32) {
      const float pj = (expf(sc[j] - L) - (j == 0 ? 1.f : 0.f)) * g;
      a += *(const f32x4*)(rw + (int64_t)j * D + k) * pj;
    }
  }
  *(f32x4*)&red[sub][l * 4] = a;
  __syncthreads();
  if (threadIdx.x < 128 && k0 + threadIdx.x < D) {
    float t = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) t += red[s2][threadIdx.x];
    df[((int64_t)side * B + b) * D + k0 + threadIdx.x] = t * invT;
  }
}

static inline hipStream_t S_(void* s) { return (hipStream_t)s; }

extern "C" int slic_nce_scores_fwd(const float* bank, const int64_t* idx, const float* f, int B, int K1, int D,
                                   float T, float* out, float* gathered, void* stream) {
  SLIC_REQUIRE(bank && idx && f && out && B > 0 && K1 > 0 && D > 0 && D % 4 == 0 && T > 0.f,
               "slic_nce_scores_fwd: bad args (D %% 4 == 0)");
  dim3 grid((unsigned)min((int)slic_cdiv(K1, 8), 64), (unsigned)B);
  nce_scores_fwd<<<grid, dim3(256), 0, S_(stream)>>>(bank, idx, f, K1, D, 1.0f / T, out, gathered);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
extern "C" int slic_nce_scores_bwd(const float* bank, const int64_t* idx, const float* dout, int B, int K1, int D,
                                   float T, float* df, void* stream) {
  SLIC_REQUIRE(bank && idx && dout && df && B > 0 && K1 > 0 && D > 0 && D % 4 == 0 && T > 0.f,
               "slic_nce_scores_bwd: bad args (D %% 4 == 0)");
  nce_scores_bwd<<<dim3(B), dim3(256), 0, S_(stream)>>>(bank, idx, dout, K1, D, 1.0f / T, df);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
extern "C" int slic_nce_bank_update(float* bank, const int64_t* y, const float* f, int B, int D, float momentum,
                                    void* stream) {
  SLIC_REQUIRE(bank && y && f && B > 0 && D > 0, "slic_nce_bank_update: bad args");
  nce_bank_update<<<dim3((unsigned)slic_cdiv(B, 4)), dim3(256), 0, S_(stream)>>>(bank, y, f, B, D, momentum);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
// loss = mean_b (logsumexp(x[b,:]) - x[b,0]); lse: [B] kept by the caller for the backward
extern "C" int slic_softmax_ce0_fwd(const float* x, int B, int K1, float* lse, float* rowloss, float* loss,
                                    void* stream) {
  SLIC_REQUIRE(x && lse && rowloss && loss && B > 0 && K1 > 0, "slic_softmax_ce0_fwd: bad args");
  softmax_ce0_fwd<<<dim3(B), dim3(256), 0, S_(stream)>>>(x, K1, lse, rowloss);
  SLIC_LAUNCH_CHECK();
  mean_rows_serial<<<dim3(1), dim3(64), 0, S_(stream)>>>(rowloss, B, loss);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
extern "C" int slic_softmax_ce0_bwd(const float* x, const float* lse, int B, int K1, const float* gscale, float* dx,
                                    void* stream) {
  SLIC_REQUIRE(x && lse && dx && B > 0 && K1 > 0, "slic_softmax_ce0_bwd: bad args");
  softmax_ce0_bwd<<<dim3((unsigned)slic_cdiv((int64_t)B * K1, 256)), dim3(256), 0, S_(stream)>>>(x, lse, B, K1, gscale, dx);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_nce_fused_fwd(const float* bank_l, const float* bank_ab, const float* f_l, const float* f_ab,
                                  const int64_t* idx, int B, int K1, int D, float T, float* scores, float* rows, float* part,
                                  void* stream) {
  static_assert(NCE_PARTS == SLIC_NCE_PARTS, "slic_hip.h: SLIC_NCE_PARTS");
  SLIC_REQUIRE(bank_l && bank_ab && f_l && f_ab && idx && scores && rows && part && B > 0 && K1 > 0 && D > 0 &&
                   D % 4 == 0 && T > 0.f, "slic_nce_fused_fwd: bad args (D %% 4 == 0)");
  // side 0 = out_ab: feature ab against memory_l; side 1 = out_l: feature l against memory_ab (NCE_loss.py:41-48)
  nce_fused_fwd<<<dim3((unsigned)B, 2, NCE_PARTS), dim3(256), 0, S_(stream)>>>(bank_l, bank_ab, f_ab, f_l, idx, K1, D, 1.0f / T, scores,
                                                                                rows, part);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
extern "C" int slic_nce_fused_update(float* bank_l, float* bank_ab, const int64_t* y, const float* f_l, const float* f_ab, int B,
                                     int D, float momentum, const float* part, const float* scores, int K1, float* lse,
                                     float* rowloss, float* loss, void* stream) {
  SLIC_REQUIRE(bank_l && bank_ab && y && f_l && f_ab && B > 0 && D > 0 && (!part || (scores && K1 > 0 && lse && rowloss)),
               "slic_nce_fused_update: bad args");
  nce_fused_update<<<dim3((unsigned)slic_cdiv(2 * B, 4)), dim3(256), 0, S_(stream)>>>(bank_l, bank_ab, y, f_l, f_ab, B, D, momentum,
                                                                                    part, scores, K1, lse, rowloss, loss);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
extern "C" int slic_nce_fused_bwd(const float* rows, const float* scores, const float* lse, int B, int K1, int D, float T,
                                  const float* gscale, float* df, void* stream) {
  SLIC_REQUIRE(rows && scores && lse && df && B > 0 && K1 > 0 && D > 0 && D % 4 == 0 && T > 0.f, "slic_nce_fused_bwd: bad args");
  nce_fused_bwd<<<dim3((unsigned)B, 2, (unsigned)slic_cdiv(D, 128)), dim3(1024), 0, S_(stream)>>>(rows, scores, lse, K1, D, 1.0f / T,
                                                                                                gscale, df);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
