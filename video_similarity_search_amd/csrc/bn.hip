// BatchNorm (train + eval), ReLU, residual add, global average pool — the HBM-bound passes between
// the conv GEMMs.  Replaces nn.BatchNorm3d / BatchNorm1d / ReLU / `out += residual` /
// AdaptiveAvgPool3d as used in /root/reference/models/resnet.py:34-57,132-133,173,183,233,294-299.
//
// Activations are [M, C] row-major (NDHWC flattened), C % 4 == 0; every pass moves 16 bytes per lane.
// Train-mode statistics are deterministic: per-workgroup partial sums (from the conv epilogue or
// bn_bwd_reduce) are added in workgroup order in double.
#include "common.h"

// Merging the per-workgroup BatchNorm partials.  A slab row r covers rows_in rows (the last one ragged) and holds
// (sum, M2 = sum (x - mean_row)^2) per channel.  One level merges groups of MG consecutive slab rows with Chan's
// pairwise update in double, in row order (deterministic); levels repeat until at most BN_MG rows are left, then
// bn_merge_final merges those and turns the result into mean / invstd / scale / shift (+ running stats).  Thread = (group, channel):
// coalesced across channels, MG serial steps — a 12544-row slab (layer1 at B = 32) takes 3 short launches
// instead of one 12544-step serial loop.
#define BN_MG 64
#define BN_MQ 4     // a group's rows are split over BN_MQ sub-chains (threadIdx.y) that run concurrently, then merged in order
template <typename Tin>
__global__ __launch_bounds__(64 * BN_MQ) void bn_merge_level(const Tin* __restrict__ in, int R, int64_t rows_in, int C,
                                                             int64_t M, double* __restrict__ out) {
  __shared__ double sh[BN_MQ][4][64];
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int q = threadIdx.y;
  const int grp = blockIdx.y;
  const int g0 = grp * BN_MG;
  const int g1 = g0 + BN_MG < R ? g0 + BN_MG : R;
  const int r0 = g0 + q * (BN_MG / BN_MQ);
  const int r1 = r0 + BN_MG / BN_MQ < g1 ? r0 + BN_MG / BN_MQ : g1;
  double n = 0.0, mean = 0.0, m2 = 0.0, sum = 0.0;
  if (c < C) {
    for (int r = r0; r < r1; ++r) {
      const int64_t left = M - (int64_t)r * rows_in;
      const double nb = (double)(left < rows_in ? left : rows_in);
      const double sb = (double)in[((int64_t)r * 2 + 0) * C + c];
      const double mb = sb / nb;
      const double d = mb - mean;
      const double nn = n + nb;
      m2 += (double)in[((int64_t)r * 2 + 1) * C + c] + d * d * n * nb / nn;
      mean += d * nb / nn;
      sum += sb;
      n = nn;
    }
  }
  sh[q][0][threadIdx.x] = n; sh[q][1][threadIdx.x] = mean; sh[q][2][threadIdx.x] = m2; sh[q][3][threadIdx.x] = sum;
  __syncthreads();
  if (q == 0 && c < C) {
#pragma unroll
    for (int u = 1; u < BN_MQ; ++u) {
      const double nb = sh[u][0][threadIdx.x];
      if (nb > 0.0) {
        const double d = sh[u][1][threadIdx.x] - mean;
        const double nn = n + nb;
        m2 += sh[u][2][threadIdx.x] + d * d * n * nb / nn;
        mean += d * nb / nn;
        sum += sh[u][3][threadIdx.x];
        n = nn;
      }
    }
    out[((int64_t)grp * 2 + 0) * C + c] = sum;
    out[((int64_t)grp * 2 + 1) * C + c] = m2;
  }
}

// Last merge level and the finalisation as ONE launch (R <= BN_MG rows left): the sub-chains of bn_merge_level, then
// scale = gamma*invstd, shift = beta - mean*scale and the running-statistics update with the UNBIASED variance (torch semantics;
// models/resnet.py uses the defaults eps = 1e-5, momentum = 0.1) — no trip through memory, no kernel boundary in between.
template <typename Tin>
__global__ __launch_bounds__(64 * BN_MQ) void bn_merge_final(const Tin* __restrict__ in, int R, int64_t rows_in, int C, int64_t M,
                                                             float eps, float momentum, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ mean_o,
                                                             float* __restrict__ invstd, float* __restrict__ scale,
                                                             float* __restrict__ shift, float* __restrict__ running_mean,
                                                             float* __restrict__ running_var) {
  __shared__ double sh[BN_MQ][4][64];
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int q = threadIdx.y;
  const int r0 = q * (BN_MG / BN_MQ);
  const int r1 = r0 + BN_MG / BN_MQ < R ? r0 + BN_MG / BN_MQ : R;
  double n = 0.0, mean = 0.0, m2 = 0.0, sum = 0.0;
  if (c < C) {
    for (int r = r0; r < r1; ++r) {
      const int64_t left = M - (int64_t)r * rows_in;
      const double nb = (double)(left < rows_in ? left : rows_in);
      const double sb = (double)in[((int64_t)r * 2 + 0) * C + c];
      const double mb = sb / nb;
      const double d = mb - mean;
      const double nn = n + nb;
      m2 += (double)in[((int64_t)r * 2 + 1) * C + c] + d * d * n * nb / nn;
      mean += d * nb / nn;
      sum += sb;
      n = nn;
    }
  }
  sh[q][0][threadIdx.x] = n; sh[q][1][threadIdx.x] = mean; sh[q][2][threadIdx.x] = m2; sh[q][3][threadIdx.x] = sum;
  __syncthreads();
  if (q == 0 && c < C) {
#pragma unroll
    for (int u = 1; u < BN_MQ; ++u) {
      const double nb = sh[u][0][threadIdx.x];
      if (nb > 0.0) {
        const double d = sh[u][1][threadIdx.x] - mean;
        const double nn = n + nb;
        m2 += sh[u][2][threadIdx.x] + d * d * n * nb / nn;
        mean += d * nb / nn;
        sum += sh[u][3][threadIdx.x];
        n = nn;
      }
    }
    // finalise from (sum, m2)
    const double mu = sum / (double)M;
    double var = m2 / (double)M;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    mean_o[c] = (float)mu;
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
}

// the backward twin: last sum level, then dgamma = s2, dbeta = s1 and ka = s1/M, kb = s2/M for pass 2
template <typename Tin>
__global__ __launch_bounds__(64 * BN_MQ) void sum_merge_final(const Tin* __restrict__ in, int R, int C, int64_t M,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              double* __restrict__ ka, double* __restrict__ kb) {
  __shared__ double sh[BN_MQ][2][64];
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int q = threadIdx.y;
  const int r0 = q * (BN_MG / BN_MQ);
  const int r1 = r0 + BN_MG / BN_MQ < R ? r0 + BN_MG / BN_MQ : R;
  double a = 0.0, b = 0.0;
  if (c < C) {
    for (int r = r0; r < r1; ++r) {
      a += (double)in[((int64_t)r * 2 + 0) * C + c];
      b += (double)in[((int64_t)r * 2 + 1) * C + c];
    }
  }
  sh[q][0][threadIdx.x] = a; sh[q][1][threadIdx.x] = b;
  __syncthreads();
  if (q == 0 && c < C) {
#pragma unroll
    for (int u = 1; u < BN_MQ; ++u) { a += sh[u][0][threadIdx.x]; b += sh[u][1][threadIdx.x]; }
    if (dbeta) dbeta[c] = (float)a;
    if (dgamma) dgamma[c] = (float)b;
    ka[c] = a / (double)M;
    kb[c] = b / (double)M;
  }
}

// plain two-plane sums in double, groups of BN_MG slab rows: BN_MQ concurrent sub-chains in row order, merged in order
template <typename Tin>
__global__ __launch_bounds__(64 * BN_MQ) void sum_merge_level(const Tin* __restrict__ in, int R, int C,
                                                              double* __restrict__ out) {
  __shared__ double sh[BN_MQ][2][64];
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int q = threadIdx.y;
  const int grp = blockIdx.y;
  const int g0 = grp * BN_MG;
  const int g1 = g0 + BN_MG < R ? g0 + BN_MG : R;
  const int r0 = g0 + q * (BN_MG / BN_MQ);
  const int r1 = r0 + BN_MG / BN_MQ < g1 ? r0 + BN_MG / BN_MQ : g1;
  double a = 0.0, b = 0.0;
  if (c < C) {
    for (int r = r0; r < r1; ++r) {
      a += (double)in[((int64_t)r * 2 + 0) * C + c];
      b += (double)in[((int64_t)r * 2 + 1) * C + c];
    }
  }
  sh[q][0][threadIdx.x] = a; sh[q][1][threadIdx.x] = b;
  __syncthreads();
  if (q == 0 && c < C) {
#pragma unroll
    for (int u = 1; u < BN_MQ; ++u) { a += sh[u][0][threadIdx.x]; b += sh[u][1][threadIdx.x]; }
    out[((int64_t)grp * 2 + 0) * C + c] = a;
    out[((int64_t)grp * 2 + 1) * C + c] = b;
  }
}

// SyncBatchNorm (online_train.py:466-468, torch.nn.SyncBatchNorm): every rank's merged (sum, M2, count) row, gathered in rank
// order, merged with Chan's update in that order (the same result on every rank) and finalised with the GLOBAL count —
// normalisation with the biased variance, running statistics with the unbiased one over all ranks' samples.
// stats: [W][2 C + 1] doubles = {sum[C], M2[C], n}.
__global__ void bn_sync_final_kernel(const double* __restrict__ stats, int W, int C, float eps, float momentum,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ mean_o,
                                     float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift,
                                     float* __restrict__ running_mean, float* __restrict__ running_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int64_t ld = 2 * (int64_t)C + 1;
  double n = 0.0, mean = 0.0, m2 = 0.0;
  for (int w = 0; w < W; ++w) {
    const double nb = stats[w * ld + 2 * C];
    if (nb > 0.0) {
      const double d = stats[w * ld + c] / nb - mean;
      const double nn = n + nb;
      m2 += stats[w * ld + C + c] + d * d * n * nb / nn;
      mean += d * nb / nn;
      n = nn;
    }
  }
  double var = n > 0.0 ? m2 / n : 0.0;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  mean_o[c] = (float)mean;
  invstd[c] = is;
  const float sc = g * is;
  scale[c] = sc;
  shift[c] = b - (float)mean * sc;
  if (running_mean) {
    const double unb = n > 1.0 ? var * (n / (n - 1.0)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
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
      // four rows' loads in flight per thread (a thread walks RB / RL rows: one 16-byte load at a time left the memory system
      // idle — 0.3-0.6 TB/s on the downsample BatchNorms); the sums stay in row order
      int64_t r = r0 + rl;
      for (; r + 3 * RL < r1; r += 4 * RL) {
        f32x4 gq[4], zq[4], oq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t e = (r + u * RL) * C4 + cg;
          gq[u] = ((const f32x4*)dy)[e];
          zq[u] = ((const f32x4*)z)[e];
          if (out) oq[u] = ((const f32x4*)out)[e];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          f32x4 g = gq[u];
          if (out) {
            g.x = oq[u].x > 0.f ? g.x : 0.f; g.y = oq[u].y > 0.f ? g.y : 0.f;
            g.z = oq[u].z > 0.f ? g.z : 0.f; g.w = oq[u].w > 0.f ? g.w : 0.f;
          }
          if (gout) ((f32x4*)gout)[(r + u * RL) * C4 + cg] = g;
          const f32x4 xh = (zq[u] - mu) * is;
          s1 += g;
          s2 += g * xh;
        }
      }
      for (; r < r1; r += RL) {
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
#ifndef SLIC_BN_DIAG
#define SLIC_BN_DIAG 0    // diagnostic builds only (scripts/r5/ab_bn_merges.sh; wrong statistics, right timing): 1 = the statistic merges are not launched
                          // (0.65-0.8 ms of a 36.7 ms step; level + final as ONE launch with a last-workgroup ticket measured equal to the two launches,
                          // 36.70 vs 36.71 ms, for the second time — round 2 — and is not kept)
#endif
static inline hipStream_t S_(void* s) { return (hipStream_t)s; }
static inline unsigned ew_grid(int64_t tot) {
  int64_t g = slic_cdiv(tot, 256);
  return (unsigned)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

// scratch for the merge levels: level l holds ceil(R / MG^l) rows of 2*C doubles (two ping-pong buffers)
static size_t merge_ws_bytes(int R, int C) {
  const size_t rows1 = (size_t)slic_cdiv(R, BN_MG);
  return 2 * slic_align_up(rows1 * 2 * C * sizeof(double), 256);
}
// runs the levels until at most BN_MG rows are left; *rows_out / *R_out / *src describe what the fused final kernel
// (bn_merge_final / sum_merge_final) still has to merge: `partial` itself (float) when R <= BN_MG, else a double buffer
template <bool CHAN>
static int run_merge(const float* partial, int R, int64_t rows, int C, int64_t M, void* ws, hipStream_t st,
                     const double** dsrc, int* R_out, int64_t* rows_out) {
  const size_t half = merge_ws_bytes(R, C) / 2;
  double* buf[2] = {(double*)ws, (double*)((char*)ws + half)};
  int cur = 0;
  int Rl = R;
  int64_t rows_l = rows;
  dim3 blk(64, BN_MQ);
  *dsrc = nullptr;
  if (Rl > BN_MG) {
    const int Ro = (int)slic_cdiv(Rl, BN_MG);
    dim3 grid((unsigned)slic_cdiv(C, 64), (unsigned)Ro);
#if !SLIC_BN_DIAG
    if (CHAN) bn_merge_level<float><<<grid, blk, 0, st>>>(partial, Rl, rows_l, C, M, buf[cur]);
    else sum_merge_level<float><<<grid, blk, 0, st>>>(partial, Rl, C, buf[cur]);
#endif
    SLIC_LAUNCH_CHECK();
    Rl = Ro;
    rows_l *= BN_MG;
    *dsrc = buf[cur];
    while (Rl > BN_MG) {
      const int Ro2 = (int)slic_cdiv(Rl, BN_MG);
      dim3 grid2((unsigned)slic_cdiv(C, 64), (unsigned)Ro2);
      if (CHAN) bn_merge_level<double><<<grid2, blk, 0, st>>>(buf[cur], Rl, rows_l, C, M, buf[cur ^ 1]);
      else sum_merge_level<double><<<grid2, blk, 0, st>>>(buf[cur], Rl, C, buf[cur ^ 1]);
      SLIC_LAUNCH_CHECK();
      cur ^= 1;
      Rl = Ro2;
      rows_l *= BN_MG;
      *dsrc = buf[cur];
    }
  }
  *R_out = Rl;
  *rows_out = rows_l;
  return SLIC_OK;
}

extern "C" size_t slic_bn_finalize_workspace_bytes(int R, int C) { return merge_ws_bytes(R, C); }

extern "C" int slic_bn_finalize(const float* partial, int R, int rows, int C, int64_t M, float eps, float momentum,
                                const float* gamma, const float* beta, float* mean, float* invstd,
                                float* scale, float* shift, float* running_mean, float* running_var,
                                void* workspace, void* stream) {
  SLIC_REQUIRE(partial && mean && invstd && scale && shift && workspace && R > 0 && C > 0 && M > 0 && rows > 0 &&
               (int64_t)R * rows >= M && (int64_t)(R - 1) * rows < M, "slic_bn_finalize: bad args (R*rows must cover M)");
  SLIC_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "slic_bn_finalize: running stats come in pairs");
  hipStream_t st = S_(stream);
  const double* dsrc = nullptr;
  int Rl = 0;
  int64_t rows_l = 0;
  int rc = run_merge<true>(partial, R, rows, C, M, workspace, st, &dsrc, &Rl, &rows_l);
  if (rc) return rc;
  const dim3 grid((unsigned)slic_cdiv(C, 64)), blk(64, BN_MQ);
#if !SLIC_BN_DIAG
  if (dsrc) bn_merge_final<double><<<grid, blk, 0, st>>>(dsrc, Rl, rows_l, C, M, eps, momentum, gamma, beta, mean, invstd, scale, shift,
                                                        running_mean, running_var);
  else bn_merge_final<float><<<grid, blk, 0, st>>>(partial, Rl, rows_l, C, M, eps, momentum, gamma, beta, mean, invstd, scale, shift,
                                                   running_mean, running_var);
#endif
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

// ---- SyncBatchNorm: the rank-local halves of slic_bn_finalize / slic_bn_bwd*, with the collective left to the caller ----
static int bwd_merge_finalize(const float* partial, int R, int64_t rows, int C, int64_t M, void* mws, hipStream_t st,
                              float* dgamma, float* dbeta, double* ka, double* kb);
// merges a conv epilogue's slab down to ONE row: stats[0 .. C) = sum, stats[C .. 2C) = M2 (doubles); the caller appends its
// sample count M as stats[2C] before the all-gather
extern "C" int slic_bn_merge_stats(const float* partial, int R, int rows, int C, int64_t M, double* stats, void* workspace,
                                   void* stream) {
  SLIC_REQUIRE(partial && stats && workspace && R > 0 && C > 0 && M > 0 && rows > 0 && (int64_t)R * rows >= M &&
               (int64_t)(R - 1) * rows < M, "slic_bn_merge_stats: bad args (R*rows must cover M)");
  hipStream_t st = S_(stream);
  const double* dsrc = nullptr;
  int Rl = 0;
  int64_t rows_l = 0;
  int rc = run_merge<true>(partial, R, rows, C, M, workspace, st, &dsrc, &Rl, &rows_l);
  if (rc) return rc;
  const dim3 grid((unsigned)slic_cdiv(C, 64), 1), blk(64, BN_MQ);
  if (dsrc) bn_merge_level<double><<<grid, blk, 0, st>>>(dsrc, Rl, rows_l, C, M, stats);
  else bn_merge_level<float><<<grid, blk, 0, st>>>(partial, Rl, rows_l, C, M, stats);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

// stats_all: [W][2 C + 1] doubles, the W ranks' slic_bn_merge_stats rows in rank order (all-gathered by the caller)
extern "C" int slic_bn_finalize_sync(const double* stats_all, int W, int C, float eps, float momentum, const float* gamma,
                                     const float* beta, float* mean, float* invstd, float* scale, float* shift,
                                     float* running_mean, float* running_var, void* stream) {
  SLIC_REQUIRE(stats_all && mean && invstd && scale && shift && W > 0 && C > 0, "slic_bn_finalize_sync: bad args");
  SLIC_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "slic_bn_finalize_sync: running stats come in pairs");
  bn_sync_final_kernel<<<dim3((unsigned)slic_cdiv(C, 64)), dim3(64), 0, S_(stream)>>>(stats_all, W, C, eps, momentum, gamma, beta, mean,
                                                                                invstd, scale, shift, running_mean, running_var);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

// backward, phase 1: the rank-local sums  sums[0 .. C) = sum g,  sums[C .. 2C) = sum g * xhat  (doubles) and the parameter
// gradients (dgamma, dbeta are rank-local, as torch's SyncBatchNorm leaves them to DistributedDataParallel).  partial = a dgrad
// epilogue's slab of R rows, or NULL: then pass 1 (ReLU mask if `out`, masked gradient to g_out if given) runs here first.
extern "C" size_t slic_bn_bwd_sums_workspace_bytes(int64_t M, int C, int R_partial) {
  const int R = R_partial > 0 ? R_partial : (int)slic_cdiv(M, BNB_RB);
  return slic_align_up((size_t)R * 2 * C * 4, 256) + slic_align_up(merge_ws_bytes(R, C), 256);
}
extern "C" int slic_bn_bwd_sums(const float* partial, int R_partial, const float* dy, const float* out, const float* z,
                                const float* mean, const float* invstd, int64_t M, int C, float* g_out, double* sums,
                                float* dgamma, float* dbeta, void* workspace, void* stream) {
  SLIC_REQUIRE(sums && workspace && M > 0 && C > 0 && C % 4 == 0, "slic_bn_bwd_sums: bad args (C %% 4 == 0)");
  SLIC_REQUIRE(partial ? R_partial > 0 : (dy && z && mean && invstd && (!out || g_out)),
               "slic_bn_bwd_sums: either a slab, or dy/z/mean/invstd (and g_out when a ReLU mask is applied)");
  hipStream_t st = S_(stream);
  SlicCarver w(workspace);
  int R = R_partial;
  if (!partial) {
    R = (int)slic_cdiv(M, BNB_RB);
    float* pl = w.take<float>((size_t)R * 2 * C);
    bn_bwd_reduce_kernel<<<dim3(R), dim3(256), 0, st>>>(dy, out, z, mean, invstd, M, C / 4, g_out, pl);
    SLIC_LAUNCH_CHECK();
    partial = pl;
  } else (void)w.take<float>((size_t)R * 2 * C);
  void* mws = w.take<char>(merge_ws_bytes(R, C));
  // sum_merge_final also writes ka = s1 / M, kb = s2 / M: pointed at `sums` with M = 1 they ARE the sums
  int rc = bwd_merge_finalize(partial, R, 1, C, 1, mws, st, dgamma, dbeta, sums, sums + C);
  return rc;
}

// backward, phase 2: dz = gamma * invstd * (g - ka - xhat * kb) with ka = global sum g / global count, kb likewise
extern "C" int slic_bn_bwd_apply(const float* g, const float* z, const float* mean, const float* invstd, const float* gamma,
                                 const double* ka, const double* kb, int64_t M, int C, float* dz, void* stream) {
  SLIC_REQUIRE(g && z && mean && invstd && gamma && ka && kb && dz && M > 0 && C > 0 && C % 4 == 0, "slic_bn_bwd_apply: bad args");
  bn_bwd_apply_kernel<<<dim3(ew_grid(M * (C / 4))), dim3(256), 0, S_(stream)>>>(g, z, mean, invstd, gamma, ka, kb, M, C / 4, dz);
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

// merge levels + the fused (last level, finalize) launch of the backward sums
static int bwd_merge_finalize(const float* partial, int R, int64_t rows, int C, int64_t M, void* mws, hipStream_t st,
                              float* dgamma, float* dbeta, double* ka, double* kb) {
  const double* dsrc = nullptr;
  int Rl = 0;
  int64_t rows_l = 0;
  int rc = run_merge<false>(partial, R, rows, C, M, mws, st, &dsrc, &Rl, &rows_l);
  if (rc) return rc;
  const dim3 grid((unsigned)slic_cdiv(C, 64)), blk(64, BN_MQ);
#if !SLIC_BN_DIAG
  if (dsrc) sum_merge_final<double><<<grid, blk, 0, st>>>(dsrc, Rl, C, M, dgamma, dbeta, ka, kb);
  else sum_merge_final<float><<<grid, blk, 0, st>>>(partial, Rl, C, M, dgamma, dbeta, ka, kb);
#endif
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_bn_bwd(const float* dy, const float* out, const float* z, const float* mean,
                           const float* invstd, const float* gamma, int64_t M, int C, float* g_out,
                           float* dz, float* dgamma, float* dbeta, void* workspace, void* stream) {
  SLIC_REQUIRE(dy && z && mean && invstd && gamma && dz && workspace && M > 0 && C > 0 && C % 4 == 0,
               "slic_bn_bwd: bad args (C %% 4 == 0)");
  hipStream_t st = S_(stream);
  const int R = (int)slic_cdiv(M, BNB_RB);
  SlicCarver w(workspace);                                   // same order as slic_bn_bwd_workspace_bytes
  float* partial = w.take<float>((size_t)R * 2 * C);
  double* ka = w.take<double>(C);
  double* kb = w.take<double>(C);
  void* mws = w.take<char>(merge_ws_bytes(R, C));
  float* gbuf = g_out;
  if (!gbuf && out) gbuf = w.take<float>((size_t)M * C);     // masked gradient must be materialised for pass 2
  bn_bwd_reduce_kernel<<<dim3(R), dim3(256), 0, st>>>(dy, out, z, mean, invstd, M, C / 4, gbuf, partial);
  SLIC_LAUNCH_CHECK();
  int rc = bwd_merge_finalize(partial, R, BNB_RB, C, M, mws, st, dgamma, dbeta, ka, kb);
  if (rc) return rc;
  bn_bwd_apply_kernel<<<dim3(ew_grid(M * (C / 4))), dim3(256), 0, st>>>(gbuf ? gbuf : dy, z, mean, invstd, gamma, ka, kb, M, C / 4, dz);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" size_t slic_bn_bwd_fused_workspace_bytes(int R, int C) {
  return 2 * slic_align_up((size_t)C * 8, 256) + slic_align_up(merge_ws_bytes(R, C), 256);
}

extern "C" int slic_bn_bwd_fused(const float* partial, int R, const float* g, const float* z, const float* mean,
                                 const float* invstd, const float* gamma, int64_t M, int C, float* dz,
                                 float* dgamma, float* dbeta, void* workspace, void* stream) {
  SLIC_REQUIRE(partial && g && z && mean && invstd && gamma && dz && workspace && R > 0 && M > 0 && C > 0 && C % 4 == 0,
               "slic_bn_bwd_fused: bad args (C %% 4 == 0)");
  hipStream_t st = S_(stream);
  SlicCarver w(workspace);                                   // same order as slic_bn_bwd_fused_workspace_bytes
  double* ka = w.take<double>(C);
  double* kb = w.take<double>(C);
  void* mws = w.take<char>(merge_ws_bytes(R, C));
  int rc = bwd_merge_finalize(partial, R, 1, C, M, mws, st, dgamma, dbeta, ka, kb);
  if (rc) return rc;
  bn_bwd_apply_kernel<<<dim3(ew_grid(M * (C / 4))), dim3(256), 0, st>>>(g, z, mean, invstd, gamma, ka, kb, M, C / 4, dz);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" size_t slic_bn_bwd_workspace_bytes(int64_t M, int C, int need_g_buffer) {
  const int R = (int)slic_cdiv(M, BNB_RB);
  size_t b = slic_align_up((size_t)R * 2 * C * 4, 256) + 2 * slic_align_up((size_t)C * 8, 256) +
             slic_align_up(merge_ws_bytes(R, C), 256);
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

// ---- the stem's MaxPool3d(kernel 3, stride 2, padding 1) (models/resnet.py:123, 262-263: `if not self.no_max_pool`), NDHWC.
// One thread per (output position, channel); window positions in (t, h, w) order, the first maximum wins (`>`; a NaN is taken and
// kept, as torch's kernel does); arg = the winning INPUT position (t * H + h) * W + w, kept for the backward.
__global__ void maxpool3d_fwd_kernel(const float* __restrict__ x, int B, int T, int H, int W, int C, int To, int Ho, int Wo,
                                     float* __restrict__ y, int32_t* __restrict__ arg) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t tot = (int64_t)B * To * Ho * Wo * C;
  if (e >= tot) return;
  const int c = (int)(e % C);
  int64_t q = e / C;
  const int wo = (int)(q % Wo); q /= Wo;
  const int ho = (int)(q % Ho); q /= Ho;
  const int to = (int)(q % To);
  const int64_t b = q / To;
  float best = -INFINITY;
  int bi = -1;
  for (int dt = 0; dt < 3; ++dt) {
    const int t = 2 * to - 1 + dt;
    if ((unsigned)t >= (unsigned)T) continue;
    for (int dh = 0; dh < 3; ++dh) {
      const int h = 2 * ho - 1 + dh;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int dw = 0; dw < 3; ++dw) {
        const int w = 2 * wo - 1 + dw;
        if ((unsigned)w >= (unsigned)W) continue;
        const int pos = (t * H + h) * W + w;
        const float v = x[((b * T * H * W) + pos) * C + c];
        if (bi < 0 || v > best || v != v) { best = v; bi = pos; }
      }
    }
  }
  y[e] = best;
  if (arg) arg[e] = bi;
}
// dx[pos] = sum of dy over the (at most eight) windows that cover pos and chose it — gather form: deterministic, no atomics
__global__ void maxpool3d_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ arg, int B, int T, int H, int W, int C,
                                     int To, int Ho, int Wo, float* __restrict__ dx) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t tot = (int64_t)B * T * H * W * C;
  if (e >= tot) return;
  const int c = (int)(e % C);
  int64_t q = e / C;
  const int pos = (int)(q % ((int64_t)T * H * W));
  const int64_t b = q / ((int64_t)T * H * W);
  const int w = pos % W, h = (pos / W) % H, t = pos / (W * H);
  float a = 0.f;
  // windows o with 2 o - 1 <= p <= 2 o + 1:  o in [ceil((p - 1) / 2), floor((p + 1) / 2)]
  for (int to = t >> 1; to <= (t + 1) >> 1; ++to) {
    if (to >= To) continue;
    for (int ho = h >> 1; ho <= (h + 1) >> 1; ++ho) {
      if (ho >= Ho) continue;
      for (int wo = w >> 1; wo <= (w + 1) >> 1; ++wo) {
        if (wo >= Wo) continue;
        const int64_t o = (((b * To + to) * Ho + ho) * Wo + wo) * C + c;
        if (arg[o] == pos) a += dy[o];
      }
    }
  }
  dx[e] = a;
}
// shortcut 'A' (models/resnet.py:213-222): F.avg_pool3d(x, kernel_size=1, stride=s) — every s-th position — then zero channels
// up to `Co`.  (The reference concatenates `out.data`: no gradient flows back through this branch, so there is no backward.)
__global__ void shortcut_a_kernel(const float* __restrict__ x, int B, int T, int H, int W, int C, int s, int To, int Ho, int Wo, int Co,
                                  float* __restrict__ y) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t tot = (int64_t)B * To * Ho * Wo * Co;
  if (e >= tot) return;
  const int c = (int)(e % Co);
  int64_t q = e / Co;
  const int wo = (int)(q % Wo); q /= Wo;
  const int ho = (int)(q % Ho); q /= Ho;
  const int to = (int)(q % To);
  const int64_t b = q / To;
  y[e] = c < C ? x[((((b * T + (int64_t)to * s) * H + (int64_t)ho * s) * W) + (int64_t)wo * s) * C + c] : 0.f;
}

extern "C" int slic_maxpool3d_fwd(const float* x, int B, int T, int H, int W, int C, float* y, int32_t* arg, void* stream) {
  SLIC_REQUIRE(x && y && B > 0 && T > 0 && H > 0 && W > 0 && C > 0 && (int64_t)T * H * W < (1ll << 31), "slic_maxpool3d_fwd: bad args");
  const int To = (T - 1) / 2 + 1, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  maxpool3d_fwd_kernel<<<dim3((unsigned)slic_cdiv((int64_t)B * To * Ho * Wo * C, 256)), dim3(256), 0, S_(stream)>>>(x, B, T, H, W, C, To, Ho, Wo, y, arg);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
extern "C" int slic_maxpool3d_bwd(const float* dy, const int32_t* arg, int B, int T, int H, int W, int C, float* dx, void* stream) {
  SLIC_REQUIRE(dy && arg && dx && B > 0 && T > 0 && H > 0 && W > 0 && C > 0, "slic_maxpool3d_bwd: bad args");
  const int To = (T - 1) / 2 + 1, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  maxpool3d_bwd_kernel<<<dim3((unsigned)slic_cdiv((int64_t)B * T * H * W * C, 256)), dim3(256), 0, S_(stream)>>>(dy, arg, B, T, H, W, C, To, Ho, Wo, dx);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
extern "C" int slic_shortcut_a(const float* x, int B, int T, int H, int W, int C, int stride, int C_out, float* y, void* stream) {
  SLIC_REQUIRE(x && y && B > 0 && T > 0 && H > 0 && W > 0 && C > 0 && stride >= 1 && C_out >= C, "slic_shortcut_a: bad args");
  const int To = (T - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  shortcut_a_kernel<<<dim3((unsigned)slic_cdiv((int64_t)B * To * Ho * Wo * C_out, 256)), dim3(256), 0, S_(stream)>>>(x, B, T, H, W, C, stride, To, Ho, Wo,
                                                                                                 C_out, y);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_colsum(const float* x, int64_t M, int C, float* out, void* stream) {
  SLIC_REQUIRE(x && out && M > 0 && C > 0, "slic_colsum: bad args");
  colsum_kernel<<<dim3((unsigned)slic_cdiv(C, 64)), dim3(64), 0, S_(stream)>>>(x, M, C, out);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
