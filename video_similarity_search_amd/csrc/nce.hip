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
