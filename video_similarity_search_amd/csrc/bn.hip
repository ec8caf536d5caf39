// BatchNorm (train + eval), ReLU, residual add, global average pool — the HBM-bound passes between
// the conv GEMMs.  Replaces nn.BatchNorm3d / BatchNorm1d / ReLU / `out += residual` /
// AdaptiveAvgPool3d as used in /root/reference/models/resnet.py:34-57,132-133,173,183,233,294-299.
//
// Activations are [M, C] row-major (NDHWC flattened), C % 4 == 0; every pass moves 16 bytes per lane.
// Train-mode statistics are deterministic: per-workgroup partial sums (from the conv epilogue or
// bn_bwd_reduce) are added in workgroup order in double.
#include "common.h"

// mean / biased var from the per-workgroup slab of (sum, M2 = sum (x - mean_blk)^2) over `rows` rows each,
// merged in workgroup order in double (Chan et al. pairwise update);
// scale = gamma*invstd, shift = beta - mean*scale; running stats: momentum update with the UNBIASED variance
// (torch semantics; models/resnet.py uses the defaults eps = 1e-5, momentum = 0.1).
__global__ void bn_finalize_kernel(const float* __restrict__ partial, int R, int rows, int C, int64_t M, float eps,
                                   float momentum, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ mean,
                                   float* __restrict__ invstd, float* __restrict__ scale,
                                   float* __restrict__ shift, float* __restrict__ running_mean,
                                   float* __restrict__ running_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int r = 0; r < R; ++r) s += (double)partial[((int64_t)r * 2 + 0) * C + c];
  const double mu = s / (double)M;
  double m2 = 0.0;
  for (int r = 0; r < R; ++r) {
    const int64_t left = M - (int64_t)r * rows;
    const double nb = (double)(left < rows ? left : rows);
    const double d = (double)partial[((int64_t)r * 2 + 0) * C + c] / nb - mu;
    m2 += (double)partial[((int64_t)r * 2 + 1) * C + c] + nb * d * d;
  }
  double var = m2 / (double)M;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  mean[c] = (float)mu;
  invstd[c] = is;
  const float sc = g * is;
  scale[c] = sc;
  shift[c] = b - (float)mu * sc;
  if (running_mean) {
    const double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

// eval mode: scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale
__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                      int C, float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.0f / sqrtf(rv[c] + eps);
  const float sc = (gamma ? gamma[c] : 1.f) * is;
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
}

// y = relu?( z*scale + shift (+ res) )
__global__ void bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                const float* __restrict__ shift, const float* __restrict__ res, int relu,
                                int64_t M, int C4, float* __restrict__ y) {
  const int64_t tot = M * C4;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C4);
    const f32x4 v = ((const f32x4*)z)[e];
    const f32x4 sc = ((const f32x4*)scale)[c];
    const f32x4 sh = ((const f32x4*)shift)[c];
    f32x4 o = v * sc + sh;
    if (res) o += ((const f32x4*)res)[e];
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    ((f32x4*)y)[e] = o;
  }
}

// Backward pass 1: g = dy * (out > 0) (if out given), written to gout (if given);
// per-workgroup partials of  s1 = sum g  and  s2 = sum g * xhat,  xhat = (z - mean) * invstd.
// Workgroup = RB rows; thread (rl, cg): channel group cg, rows rl, rl + RL, ...
#define BNB_RB 256
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const float* __restrict__ dy, const float* __restrict__ out, const float* __restrict__ z,
    const float* __restrict__ mean, const float* __restrict__ invstd, int64_t M, int C4,
    float* __restrict__ gout, float* __restrict__ partial /* [nblk][2][C] */) {
  __shared__ f32x4 red[2][256];
  const int CG = C4 < 256 ? C4 : 256;
  const int RL = 256 / CG;
  const int t = threadIdx.x;
  const int cgl = t % CG, rl = t / CG;
  const int64_t r0 = (int64_t)blockIdx.x * BNB_RB;
  const int64_t r1 = r0 + BNB_RB < M ? r0 + BNB_RB : M;
  const int C = C4 * 4;
  for (int cb = 0; cb < C4; cb += CG) {
    const int cg = cb + cgl;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (cg < C4 && rl < RL) {
      const f32x4 mu = ((const f32x4*)mean)[cg];
      const f32x4 is = ((const f32x4*)invstd)[cg];
      for (int64_t r = r0 + rl; r < r1; r += RL) {
        const int64_t e = r * C4 + cg;
        f32x4 g = ((const f32x4*)dy)[e];
        if (out) {
          const f32x4 o = ((const f32x4*)out)[e];
          g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f;
          g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
        }
        if (gout) ((f32x4*)gout)[e] = g;
        const f32x4 xh = (((const f32x4*)z)[e] - mu) * is;
        s1 += g;
        s2 += g * xh;
      }
    }
    red[0][t] = s1;
    red[1][t] = s2;
    __syncthreads();
    if (t < CG && cb + t < C4) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
      for (int u = 0; u < RL; ++u) { a += red[0][u * CG + t]; b += red[1][u * CG + t]; }
      *(f32x4*)&partial[((int64_t)blockIdx.x * 2 + 0) * C + (cb + t) * 4] = a;
      *(f32x4*)&partial[((int64_t)blockIdx.x * 2 + 1) * C + (cb + t) * 4] = b;
    }
    __syncthreads();
  }
}

// dgamma = s2, dbeta = s1 (sums of the partial slab, workgroup order, double);
// ka = s1/M, kb = s2/M for pass 2
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partial, int R, int C, int64_t M,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       double* __restrict__ ka, double* __restrict__ kb) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int r = 0; r < R; ++r) {
    s1 += (double)partial[((int64_t)r * 2 + 0) * C + c];
    s2 += (double)partial[((int64_t)r * 2 + 1) * C + c];
  }
  if (dbeta) dbeta[c] = (float)s1;
  if (dgamma) dgamma[c] = (float)s2;
  ka[c] = s1 / (double)M;
  kb[c] = s2 / (double)M;
}

// Backward pass 2: dz = gamma*invstd * (g - ka - xhat*kb).  The bracket cancels to ~(1 - xhat^2) of its terms
// when a channel has few samples (BatchNorm1d over a small batch), so it is evaluated in double, like
// torch's CPU kernel (accscalar = double), with the same fp32 xhat that pass 1 summed.
__global__ void bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ z,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const double* __restrict__ ka,
                                    const double* __restrict__ kb, int64_t M, int C4, float* __restrict__ dz) {
  const int64_t tot = M * C4;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C4);
    const f32x4 mu = ((const f32x4*)mean)[c], is = ((const f32x4*)invstd)[c];
    const f32x4 gm = ((const f32x4*)gamma)[c];
    const f32x4 xh = (((const f32x4*)z)[e] - mu) * is;
    const f32x4 gv = ((const f32x4*)g)[e];
    f32x4 o;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double br = (double)gv[u] - ka[c * 4 + u] - (double)xh[u] * kb[c * 4 + u];
      o[u] = (float)((double)gm[u] * (double)is[u] * br);
    }
    ((f32x4*)dz)[e] = o;
  }
}

// y[b, c] = mean_s x[b, s, c]
__global__ void avgpool_fwd_kernel(const float* __restrict__ x, int B, int S, int C4, float* __restrict__ y) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * C4) return;
  const int b = e / C4, c = e % C4;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < S; ++s) a += ((const f32x4*)x)[((int64_t)b * S + s) * C4 + c];
  const float inv = 1.0f / (float)S;
  ((f32x4*)y)[e] = a * inv;
}
// dx[b, s, c] = dy[b, c] / S
__global__ void avgpool_bwd_kernel(const float* __restrict__ dy, int B, int S, int C4, float* __restrict__ dx) {
  const int64_t tot = (int64_t)B * S * C4;
  const float inv = 1.0f / (float)S;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C4);
    const int64_t b = e / ((int64_t)S * C4);
    ((f32x4*)dx)[e] = ((const f32x4*)dy)[b * C4 + c] * inv;
  }
}

// out[c] = sum_m x[m, c] in double, rows ascending (bias gradients; M is a batch size here)
__global__ void colsum_kernel(const float* __restrict__ x, int64_t M, int C, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0;
  for (int64_t m = 0; m < M; ++m) a += (double)x[m * C + c];
  out[c] = (float)a;
}

// ------------------------------------ C ABI ------------------------------------------------
static inline hipStream_t S_(void* s) { return (hipStream_t)s; }
static inline unsigned ew_grid(int64_t tot) {
  int64_t g = slic_cdiv(tot, 256);
  return (unsigned)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

extern "C" int slic_bn_finalize(const float* partial, int R, int rows, int C, int64_t M, float eps, float momentum,
                                const float* gamma, const float* beta, float* mean, float* invstd,
                                float* scale, float* shift, float* running_mean, float* running_var,
                                void* stream) {
  SLIC_REQUIRE(partial && mean && invstd && scale && shift && R > 0 && C > 0 && M > 0 && rows > 0 &&
               (int64_t)R * rows >= M && (int64_t)(R - 1) * rows < M, "slic_bn_finalize: bad args (R*rows must cover M)");
  SLIC_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "slic_bn_finalize: running stats come in pairs");
  bn_finalize_kernel<<<dim3((unsigned)slic_cdiv(C, 64)), dim3(64), 0, S_(stream)>>>(
      partial, R, rows, C, M, eps, momentum, gamma, beta, mean, invstd, scale, shift, running_mean, running_var);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, int C, float* scale, float* shift,
                                   void* stream) {
  SLIC_REQUIRE(running_mean && running_var && scale && shift && C > 0, "slic_bn_eval_affine: bad args");
  bn_eval_affine_kernel<<<dim3((unsigned)slic_cdiv(C, 64)), dim3(64), 0, S_(stream)>>>(gamma, beta, running_mean, running_var, eps, C, scale, shift);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_bn_apply(const float* z, const float* scale, const float* shift, const float* res,
                             int relu, int64_t M, int C, float* y, void* stream) {
  SLIC_REQUIRE(z && scale && shift && y && M > 0 && C > 0 && C % 4 == 0, "slic_bn_apply: bad args (C %% 4 == 0)");
  bn_apply_kernel<<<dim3(ew_grid(M * (C / 4))), dim3(256), 0, S_(stream)>>>(z, scale, shift, res, relu, M, C / 4, y);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_bn_bwd_rows_per_partial(void) { return BNB_RB; }

extern "C" int slic_bn_bwd(const float* dy, const float* out, const float* z, const float* mean,
                           const float* invstd, const float* gamma, int64_t M, int C, float* g_out,
                           float* dz, float* dgamma, float* dbeta, void* workspace, void* stream) {
  SLIC_REQUIRE(dy && z && mean && invstd && gamma && dz && workspace && M > 0 && C > 0 && C % 4 == 0,
               "slic_bn_bwd: bad args (C %% 4 == 0)");
  hipStream_t st = S_(stream);
  const int R = (int)slic_cdiv(M, BNB_RB);
  SlicCarver w(workspace);
  float* partial = w.take<float>((size_t)R * 2 * C);
  double* ka = w.take<double>(C);
  double* kb = w.take<double>(C);
  float* gbuf = g_out;
  if (!gbuf && out) gbuf = w.take<float>((size_t)M * C);   // masked gradient must be materialised for pass 2
  bn_bwd_reduce_kernel<<<dim3(R), dim3(256), 0, st>>>(dy, out, z, mean, invstd, M, C / 4, gbuf, partial);
  SLIC_LAUNCH_CHECK();
  bn_bwd_finalize_kernel<<<dim3((unsigned)slic_cdiv(C, 64)), dim3(64), 0, st>>>(partial, R, C, M, dgamma, dbeta, ka, kb);
  SLIC_LAUNCH_CHECK();
  bn_bwd_apply_kernel<<<dim3(ew_grid(M * (C / 4))), dim3(256), 0, st>>>(gbuf ? gbuf : dy, z, mean, invstd, gamma, ka, kb, M, C / 4, dz);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" size_t slic_bn_bwd_workspace_bytes(int64_t M, int C, int need_g_buffer) {
  size_t b = slic_align_up((size_t)slic_cdiv(M, BNB_RB) * 2 * C * 4, 256) + 2 * slic_align_up((size_t)C * 8, 256);
  if (need_g_buffer) b += slic_align_up((size_t)M * C * 4, 256);
  return b;
}

extern "C" int slic_avgpool_fwd(const float* x, int B, int S, int C, float* y, void* stream) {
  SLIC_REQUIRE(x && y && B > 0 && S > 0 && C > 0 && C % 4 == 0, "slic_avgpool_fwd: bad args");
  avgpool_fwd_kernel<<<dim3((unsigned)slic_cdiv((int64_t)B * C / 4, 64)), dim3(64), 0, S_(stream)>>>(x, B, S, C / 4, y);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
extern "C" int slic_avgpool_bwd(const float* dy, int B, int S, int C, float* dx, void* stream) {
  SLIC_REQUIRE(dy && dx && B > 0 && S > 0 && C > 0 && C % 4 == 0, "slic_avgpool_bwd: bad args");
  avgpool_bwd_kernel<<<dim3(ew_grid((int64_t)B * S * C / 4)), dim3(256), 0, S_(stream)>>>(dy, B, S, C / 4, dx);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_colsum(const float* x, int64_t M, int C, float* out, void* stream) {
  SLIC_REQUIRE(x && out && M > 0 && C > 0, "slic_colsum: bad args");
  colsum_kernel<<<dim3((unsigned)slic_cdiv(C, 64)), dim3(64), 0, S_(stream)>>>(x, M, C, out);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
