// InfoNCE / NT-Xent pairwise-similarity loss, fused.
//
// Replaces OnlineTripletLoss.forward(..., sampling_strategy='noise_contrastive') of
// /root/reference/loss/triplet_loss.py:95-116 with pdist (:429-437): a Python loop of 2B
// F.cosine_similarity launches + masked_fill + F.cross_entropy becomes
//   ntxent_normalize  : e_hat = e / max(||e||, 1e-8)            (F.cosine_similarity's per-norm clamp)
//   ntxent_fwd        : S = e_hat e_hat^T on fp32 MFMA, sim = 1 - (1 - S), diag := 0 (NOT -inf: exp(0)
//                       stays in the denominator, :101), / T, row log-sum-exp, target (n/2 + i) mod n (:106-109)
//   ntxent_bwd        : gradient to all n embeddings, through the normalisation.
// MFMA roles are swapped (A = column block j, B = row block i) so a lane owns ONE row i and sees 16 j's
// per accumulator: the softmax reduction stays in registers, then one shuffle + one LDS step.
#include "common.h"
#include <math.h>

__global__ void ntxent_normalize(const float* __restrict__ E, int n, int D, int lde,
                                 float* __restrict__ Eh, float* __restrict__ rnorm) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  const float* e = E + (int64_t)row * lde;
  float s = 0.f;
  for (int k = lane; k < D; k += 64) s = fmaf(e[k], e[k], s);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float nrm = fmaxf(sqrtf(s), 1e-8f);
  const float rn = 1.0f / nrm;
  if (lane == 0) rnorm[row] = rn;
  for (int k = lane; k < D; k += 64) Eh[(int64_t)row * D + k] = e[k] / nrm;
}

// workgroup = 32 rows i; wave w walks column tiles w, w+4, ...
__global__ __launch_bounds__(256) void ntxent_fwd_kernel(const float* __restrict__ Eh, int n, int D, float invT,
                                                         float* __restrict__ lse,
                                                         float* __restrict__ rowloss) {
  __shared__ float wsum[4][32], wtgt[4][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int i0 = blockIdx.x * 32;
  const int i = i0 + r;
  const int ic = i < n ? i : n - 1;
  const int tgt = (n / 2 + i) % n;
  const float m = invT;  // sim/T <= 1/T: fixed shift for the log-sum-exp
  float sum = 0.f, st = 0.f;
  const int ntile = (n + 31) / 32;
  for (int jt = wave; jt < ntile; jt += 4) {
    const int j0 = jt * 32;
    const int jr = (j0 + r) < n ? (j0 + r) : n - 1;
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = 0.f;
    const float* pa = Eh + (int64_t)jr * D + h;   // A: rows = j
    const float* pb = Eh + (int64_t)ic * D + h;   // B: cols = i
    for (int k = 0; k < D; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k], pb[k], acc, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int j = j0 + (g & 3) + 8 * (g >> 2) + 4 * h;
      if (j < n) {
        float s = 1.0f - (1.0f - acc[g]);   // sim_matrix = 1 - pdist (:100)
        if (j == i) s = 0.f;                // masked_fill_(eye, 0) (:101)
        s *= invT;
        sum += expf(s - m);
        if (j == tgt) st = s;
      }
    }
  }
  sum += __shfl_xor(sum, 32);
  st += __shfl_xor(st, 32);
  if (h == 0) { wsum[wave][r] = sum; wtgt[wave][r] = st; }
  __syncthreads();
  if (threadIdx.x < 32 && i0 + threadIdx.x < n) {
    const int t = threadIdx.x;
    const float s = wsum[0][t] + wsum[1][t] + wsum[2][t] + wsum[3][t];
    const float tg = wtgt[0][t] + wtgt[1][t] + wtgt[2][t] + wtgt[3][t];
    const float l = m + logf(s);
    lse[i0 + t] = l;
    rowloss[i0 + t] = l - tg;
  }
}

__global__ void mean_serial(const float* __restrict__ v, int n, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double a = 0.0;
    for (int i = 0; i < n; ++i) a += (double)v[i];
    *out = (float)(a / (double)n);
  }
}

// workgroup = row i.  H_ij = (p_ij - [j == t(i)]) + (p_ji - [i == t(j)]), zero on the diagonal;
// d e_hat_i = (gscale / (n T)) * sum_j H_ij e_hat_j ;  d e_i = (d e_hat_i - e_hat_i <e_hat_i, d e_hat_i>) * rnorm_i
__global__ __launch_bounds__(256) void ntxent_bwd_kernel(const float* __restrict__ Eh,
                                                         const float* __restrict__ rnorm,
                                                         const float* __restrict__ lse, int n, int D,
                                                         float invT, const float* __restrict__ gscale,
                                                         float* __restrict__ dE, int ldd) {
  extern __shared__ float sm[];  // [n] H row, [D] d e_hat, [4] reduce
  float* Hrow = sm;
  float* dh = sm + n;
  float* red = sm + n + D;
  const int i = blockIdx.x, t = threadIdx.x;
  const float* ei = Eh + (int64_t)i * D;
  const int ti = (n / 2 + i) % n;
  const float li = lse[i];
  for (int j = t; j < n; j += 256) {
    float hv = 0.f;
    if (j != i) {
      const float* ej = Eh + (int64_t)j * D;
      float c = 0.f;
      for (int k = 0; k < D; ++k) c = fmaf(ei[k], ej[k], c);
      const float s = (1.0f - (1.0f - c)) * invT;
      const int tj = (n / 2 + j) % n;
      hv = (expf(s - li) - (j == ti ? 1.f : 0.f)) + (expf(s - lse[j]) - (i == tj ? 1.f : 0.f));
    }
    Hrow[j] = hv;
  }
  __syncthreads();
  const float gs = (gscale ? *gscale : 1.0f) * invT / (float)n;
  float part = 0.f;
  for (int k = t; k < D; k += 256) {
    float a = 0.f;
    for (int j = 0; j < n; ++j) a = fmaf(Hrow[j], Eh[(int64_t)j * D + k], a);
    a *= gs;
    dh[k] = a;
    part = fmaf(a, ei[k], part);
  }
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
  if ((t & 63) == 0) red[t >> 6] = part;
  __syncthreads();
  const float dot = red[0] + red[1] + red[2] + red[3];
  const float rn = rnorm[i];
  for (int k = t; k < D; k += 256) dE[(int64_t)i * ldd + k] = (dh[k] - ei[k] * dot) * rn;
}

// one wave per row: 1 - cos(x, y) with F.cosine_similarity's per-norm clamp, or ||x - y + 1e-6||_2
// (F.pairwise_distance default eps) — models/triplet_net.py:29-33
__global__ void pair_distance_kernel(const float* __restrict__ X, const float* __restrict__ Y, int n, int D,
                                     int euclid, float* __restrict__ out) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  const float* x = X + (int64_t)row * D;
  const float* y = Y + (int64_t)row * D;
  float a = 0.f, b = 0.f, c = 0.f;
  for (int k = lane; k < D; k += 64) {
    if (euclid) { const float d = x[k] - y[k] + 1e-6f; a = fmaf(d, d, a); }
    else { a = fmaf(x[k], y[k], a); b = fmaf(x[k], x[k], b); c = fmaf(y[k], y[k], c); }
  }
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
  if (lane == 0) out[row] = euclid ? sqrtf(a) : 1.0f - a / (fmaxf(sqrtf(b), 1e-8f) * fmaxf(sqrtf(c), 1e-8f));
}

// backward of pair_distance_kernel: dX, dY from g = dL/d dist (one wave per row).
//   cosine:    d = 1 - s / (nx ny), nx = max(|x|, 1e-8), ny likewise: dd/dx = -(y / (nx ny) - s x / (nx^3 ny)) where |x| > 1e-8
//              (the clamp is flat below it: only the y / (nx ny) term is left), symmetrically for y
//   euclidean: d = |x - y + 1e-6|: dd/dx = (x - y + 1e-6) / d = -dd/dy  (0 where d == 0, as torch's norm backward)
__global__ void pair_distance_bwd_kernel(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ G, int n,
                                         int D, int euclid, float* __restrict__ dX, float* __restrict__ dY) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  const float* x = X + (int64_t)row * D;
  const float* y = Y + (int64_t)row * D;
  float a = 0.f, b = 0.f, c = 0.f;
  for (int k = lane; k < D; k += 64) {
    if (euclid) { const float d = x[k] - y[k] + 1e-6f; a = fmaf(d, d, a); }
    else { a = fmaf(x[k], y[k], a); b = fmaf(x[k], x[k], b); c = fmaf(y[k], y[k], c); }
  }
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
  const float g = G[row];
  float* dx = dX + (int64_t)row * D;
  float* dy = dY + (int64_t)row * D;
  if (euclid) {
    const float d = sqrtf(a), inv = d > 0.f ? g / d : 0.f;
    for (int k = lane; k < D; k += 64) { const float v = (x[k] - y[k] + 1e-6f) * inv; dx[k] = v; dy[k] = -v; }
    return;
  }
  const float rx = sqrtf(b), ry = sqrtf(c);
  const float nx = fmaxf(rx, 1e-8f), ny = fmaxf(ry, 1e-8f);
  const float inv = 1.0f / (nx * ny);
  const float cx = rx > 1e-8f ? a * inv / (nx * nx) : 0.f;      // s / (nx^3 ny)
  const float cy = ry > 1e-8f ? a * inv / (ny * ny) : 0.f;
  for (int k = lane; k < D; k += 64) {
    dx[k] = -g * (y[k] * inv - cx * x[k]);
    dy[k] = -g * (x[k] * inv - cy * y[k]);
  }
}

// full distance matrix (loss/triplet_loss.py:429-437): one wave per (i, j)
__global__ void pdist_kernel(const float* __restrict__ V, int n, int D, float eps, int euclid,
                             float* __restrict__ out) {
  const int64_t pair = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (pair >= (int64_t)n * n) return;
  const int i = (int)(pair / n), j = (int)(pair % n);
  const int lane = threadIdx.x & 63;
  const float* x = V + (int64_t)i * D;
  const float* y = V + (int64_t)j * D;
  float a = 0.f, b = 0.f, c = 0.f;
  for (int k = lane; k < D; k += 64) {
    if (euclid) { const float d = x[k] - y[k] + eps; a = fmaf(d, d, a); }
    else { a = fmaf(x[k], y[k], a); b = fmaf(x[k], x[k], b); c = fmaf(y[k], y[k], c); }
  }
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
  if (lane == 0) out[pair] = euclid ? sqrtf(a) : 1.0f - a / (fmaxf(sqrtf(b), 1e-8f) * fmaxf(sqrtf(c), 1e-8f));
}

// rectangular distance matrix (loss/triplet_loss.py:439-447 pdist_v2): one wave per (i, j)
__global__ void pdist2_kernel(const float* __restrict__ X, int nx, const float* __restrict__ Y, int ny, int D, float eps,
                              int euclid, float* __restrict__ out) {
  const int64_t pair = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (pair >= (int64_t)nx * ny) return;
  const int i = (int)(pair / ny), j = (int)(pair % ny);
  const int lane = threadIdx.x & 63;
  const float* x = X + (int64_t)i * D;
  const float* y = Y + (int64_t)j * D;
  float a = 0.f, b = 0.f, c = 0.f;
  for (int k = lane; k < D; k += 64) {
    if (euclid) { const float d = x[k] - y[k] + eps; a = fmaf(d, d, a); }
    else { a = fmaf(x[k], y[k], a); b = fmaf(x[k], x[k], b); c = fmaf(y[k], y[k], c); }
  }
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
  if (lane == 0) out[pair] = euclid ? sqrtf(a) : 1.0f - a / (fmaxf(sqrtf(b), 1e-8f) * fmaxf(sqrtf(c), 1e-8f));
}

// InfoNCE over gathered rows — the 'all_semi_hard' branch of OnlineTripletLoss (loss/triplet_loss.py:118-203): per anchor row x
// with NY rows y_0 (the positive), y_1 .. y_{NY-1} (the picked negatives):
//   d_j = 1 - cos(x, y_j),  sim_j = exp((1 - d_j) / T),  loss = -log(sim_0 / (sum_{j >= 1} sim_j + sim_0)),  mean over the rows.
// One wave per anchor row; the cosines and norms are kept for the backward.  NY <= 8.
#define SLIC_INFONCE_MAXY 8
__global__ void infonce_rows_fwd_kernel(const float* __restrict__ X, const float* __restrict__ Y, int P, int NY, int D,
                                        float inv_t, float* __restrict__ st /* [P][2 * MAXY + 2] */, float* __restrict__ rowloss) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= P) return;
  const int lane = threadIdx.x & 63;
  const float* x = X + (int64_t)row * D;
  const float* y = Y + (int64_t)row * NY * D;
  float xy[SLIC_INFONCE_MAXY], yy[SLIC_INFONCE_MAXY], xx = 0.f;
#pragma unroll
  for (int j = 0; j < SLIC_INFONCE_MAXY; ++j) { xy[j] = 0.f; yy[j] = 0.f; }
  for (int k = lane; k < D; k += 64) {
    const float xv = x[k];
    xx = fmaf(xv, xv, xx);
#pragma unroll
    for (int j = 0; j < SLIC_INFONCE_MAXY; ++j)
      if (j < NY) { const float yv = y[(int64_t)j * D + k]; xy[j] = fmaf(xv, yv, xy[j]); yy[j] = fmaf(yv, yv, yy[j]); }
  }
  for (int o = 32; o > 0; o >>= 1) {
    xx += __shfl_xor(xx, o);
#pragma unroll
    for (int j = 0; j < SLIC_INFONCE_MAXY; ++j) { xy[j] += __shfl_xor(xy[j], o); yy[j] += __shfl_xor(yy[j], o); }
  }
  if (lane == 0) {
    float* s = st + (int64_t)row * (2 * SLIC_INFONCE_MAXY + 2);
    const float nx = fmaxf(sqrtf(xx), 1e-8f);
    s[2 * SLIC_INFONCE_MAXY] = nx;
    float sim0 = 0.f, an = 0.f;
#pragma unroll
    for (int j = 0; j < SLIC_INFONCE_MAXY; ++j)
      if (j < NY) {
        const float ny = fmaxf(sqrtf(yy[j]), 1e-8f);
        const float c = xy[j] / (nx * ny);
        const float d = 1.f - c;                        // the reference exponentiates (1 - dist) / T with dist = 1 - cos
        const float e = expf((1.f - d) * inv_t);
        s[j] = c; s[SLIC_INFONCE_MAXY + j] = ny;
        if (j == 0) sim0 = e; else an += e;
      }
    s[2 * SLIC_INFONCE_MAXY + 1] = an + sim0;
    rowloss[row] = -logf(sim0 / (an + sim0));
  }
}
// d loss / d s_0 = -(1 - p_0), d loss / d s_j = p_j (p = softmax over the NY similarities s_j = cos_j / T); d cos(x,y)/dx as above
__global__ void infonce_rows_bwd_kernel(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ st,
                                        int P, int NY, int D, float inv_t, const float* __restrict__ gscale,
                                        float* __restrict__ dX, float* __restrict__ dY) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= P) return;
  const int lane = threadIdx.x & 63;
  const float* s = st + (int64_t)row * (2 * SLIC_INFONCE_MAXY + 2);
  const float nx = s[2 * SLIC_INFONCE_MAXY], Z = s[2 * SLIC_INFONCE_MAXY + 1];
  const float g = (gscale ? *gscale : 1.f) / (float)P;
  float w[SLIC_INFONCE_MAXY], c[SLIC_INFONCE_MAXY], ny[SLIC_INFONCE_MAXY];     // w_j = g * d loss / d cos_j
#pragma unroll
  for (int j = 0; j < SLIC_INFONCE_MAXY; ++j) {
    c[j] = j < NY ? s[j] : 0.f;
    ny[j] = j < NY ? s[SLIC_INFONCE_MAXY + j] : 1.f;
    const float pj = j < NY ? expf((1.f - (1.f - c[j])) * inv_t) / Z : 0.f;
    w[j] = g * inv_t * (j == 0 ? pj - 1.f : pj);
  }
  const float* x = X + (int64_t)row * D;
  const float* y = Y + (int64_t)row * NY * D;
  for (int k = lane; k < D; k += 64) {
    const float xv = x[k];
    float dx = 0.f;
#pragma unroll
    for (int j = 0; j < SLIC_INFONCE_MAXY; ++j)
      if (j < NY) {
        const float yv = y[(int64_t)j * D + k];
        dx += w[j] * (yv / (nx * ny[j]) - c[j] * xv / (nx * nx));
        dY[((int64_t)row * NY + j) * D + k] = w[j] * (xv / (nx * ny[j]) - c[j] * yv / (ny[j] * ny[j]));
      }
    dX[(int64_t)row * D + k] = dx;
  }
}

// LLC / relative-speed margin term of triplet_train_epoch (online_train.py:317-332):
//   d1 = 1 - cos(x, y), d2 = 1 - cos(x, z), MarginRankingLoss(margin)(d1, d2, target = -1) = mean(max(0, d1 - d2 + margin)).
// One wave per row; row state (dots and norms) is kept for the backward.
__global__ void margin_cos_fwd_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                      const float* __restrict__ Z, int n, int D, float margin,
                                      float* __restrict__ st /* [n][8] */, float* __restrict__ rowloss) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  const float *x = X + (int64_t)row * D, *y = Y + (int64_t)row * D, *z = Z + (int64_t)row * D;
  float xy = 0.f, xz = 0.f, xx = 0.f, yy = 0.f, zz = 0.f;
  for (int k = lane; k < D; k += 64) {
    xy = fmaf(x[k], y[k], xy); xz = fmaf(x[k], z[k], xz);
    xx = fmaf(x[k], x[k], xx); yy = fmaf(y[k], y[k], yy); zz = fmaf(z[k], z[k], zz);
  }
  for (int o = 32; o > 0; o >>= 1) {
    xy += __shfl_xor(xy, o); xz += __shfl_xor(xz, o); xx += __shfl_xor(xx, o); yy += __shfl_xor(yy, o); zz += __shfl_xor(zz, o);
  }
  if (lane == 0) {
    const float nx = fmaxf(sqrtf(xx), 1e-8f), ny = fmaxf(sqrtf(yy), 1e-8f), nz = fmaxf(sqrtf(zz), 1e-8f);
    const float c1 = xy / (nx * ny), c2 = xz / (nx * nz);
    const float v = (1.f - c1) - (1.f - c2) + margin;
    float* s = st + (int64_t)row * 8;
    s[0] = c1; s[1] = c2; s[2] = nx; s[3] = ny; s[4] = nz; s[5] = v > 0.f ? 1.f : 0.f;
    rowloss[row] = fmaxf(v, 0.f);
  }
}
// d/dx [-(c1) + c2] etc.:  d cos(x,y)/dx = y/(|x||y|) - c x/|x|^2
__global__ void margin_cos_bwd_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                      const float* __restrict__ Z, const float* __restrict__ st, int n, int D,
                                      const float* __restrict__ gscale, float* __restrict__ dX,
                                      float* __restrict__ dY, float* __restrict__ dZ) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  const float* s = st + (int64_t)row * 8;
  const float c1 = s[0], c2 = s[1], nx = s[2], ny = s[3], nz = s[4];
  const float g = s[5] * (gscale ? *gscale : 1.f) / (float)n;     // d loss / d v
  const float *x = X + (int64_t)row * D, *y = Y + (int64_t)row * D, *z = Z + (int64_t)row * D;
  for (int k = lane; k < D; k += 64) {
    const float dc1_dx = y[k] / (nx * ny) - c1 * x[k] / (nx * nx);
    const float dc2_dx = z[k] / (nx * nz) - c2 * x[k] / (nx * nx);
    dX[(int64_t)row * D + k] = g * (-dc1_dx + dc2_dx);
    dY[(int64_t)row * D + k] = g * -(x[k] / (nx * ny) - c1 * y[k] / (ny * ny));
    dZ[(int64_t)row * D + k] = g * (x[k] / (nx * nz) - c2 * z[k] / (nz * nz));
  }
}

// Negative selection of NegativeTripletSelector.get_one_one_triplets (loss/triplet_loss.py:311-360): one wave per
// (anchor, positive) pair over the [n, n] distance matrix.
//   mode 0 'random_negative'  : uniform over the rows with a different label            (:332-333)
//   mode 1 'random_semi_hard' : uniform over {j : d(a,p) + margin - d(a,j) > 0}         (:334-336, 368-377)
//   mode 2 'fixed_semi_hard'  : argmax_j d(a,p) + margin - d(a,j) if that set is non-empty (:342-344, 398-406)
// u[pair] in [0, 1) replaces Python's random.choice (rank floor(u * count) in ascending index order).
// No semi-hard/hard negative -> hardest_easy_sampling (:350-351, 424-426): argmin of d(a, negatives) — and, exactly
// like the reference, the POSITION in the negatives list is returned, not the row it stands for.
//   mode 3 'adapted_hard'     : MemTripletLoss's default — the reference's adapted_hard_sampling has no return statement
//                               (:408-421), so every pair takes the hardest-easy fallback
// The matrix may be rectangular (MemTripletLoss: rows = the batch, columns = the queue): `labels` are the COLUMN labels, n the
// column count = row stride, and the anchor's label comes from anc_label[pair] when given (else labels[anchor row]).
__global__ void triplet_select_kernel(const float* __restrict__ Dm, const int64_t* __restrict__ labels, int n,
                                      const int32_t* __restrict__ anc, const int64_t* __restrict__ anc_label,
                                      const int32_t* __restrict__ pos, int P,
                                      float margin, int mode, const float* __restrict__ u,
                                      int32_t* __restrict__ neg) {
  const int pr = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (pr >= P) return;
  const int lane = threadIdx.x & 63;
  const int a = anc[pr];
  const int64_t la = anc_label ? anc_label[pr] : labels[a];
  const float thr = Dm[(int64_t)a * n + pos[pr]] + margin;
  const float* row = Dm + (int64_t)a * n;
  // pass 1: counts, argmax of loss / argmin of distance over the negatives (first index wins)
  int cnt_neg = 0, cnt_c = 0;
  float best_loss = -INFINITY, best_d = INFINITY;
  int best_loss_j = -1, best_d_j = -1;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    const bool isneg = j < n && labels[j] != la;
    const float d = isneg ? row[j] : 0.f;
    const bool c = isneg && (thr - d > 0.f);
    cnt_neg += __popcll(__ballot(isneg));
    cnt_c += __popcll(__ballot(c));
    if (isneg) {
      if (thr - d > best_loss) { best_loss = thr - d; best_loss_j = j; }
      if (d < best_d) { best_d = d; best_d_j = j; }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float ol = __shfl_xor(best_loss, o); const int oj = __shfl_xor(best_loss_j, o);
    if (oj >= 0 && (ol > best_loss || (ol == best_loss && (best_loss_j < 0 || oj < best_loss_j)))) { best_loss = ol; best_loss_j = oj; }
    const float od = __shfl_xor(best_d, o); const int odj = __shfl_xor(best_d_j, o);
    if (odj >= 0 && (od < best_d || (od == best_d && (best_d_j < 0 || odj < best_d_j)))) { best_d = od; best_d_j = odj; }
  }
  int result = -1;
  const int want_set = mode == 0 ? cnt_neg : cnt_c;       // size of the set a rank is drawn from
  if (mode == 3) {
    // straight to the fallback
  } else if (mode == 2) {
    if (cnt_c > 0) result = best_loss_j;
  } else if (want_set > 0) {
    int rsel = (int)(u[pr] * (float)want_set);
    if (rsel >= want_set) rsel = want_set - 1;
    int seen = 0;
    for (int j0 = 0; j0 < n && result < 0; j0 += 64) {
      const int j = j0 + lane;
      const bool isneg = j < n && labels[j] != la;
      const bool c = mode == 0 ? isneg : (isneg && (thr - row[j] > 0.f));
      const unsigned long long m = __ballot(c);
      const int pc = __popcll(m);
      if (seen + pc > rsel) {
        unsigned long long mm = m;
        for (int k = 0; k < rsel - seen; ++k) mm &= mm - 1;      // drop the lowest (rsel - seen) set bits
        result = j0 + (__ffsll((long long)mm) - 1);
      }
      seen += pc;
    }
  }
  if (result < 0) {
    // hardest_easy_sampling: position of the closest negative inside the negatives list (reference quirk)
    int rank = 0;
    for (int j0 = 0; j0 < best_d_j; j0 += 64) {
      const int j = j0 + lane;
      rank += __popcll(__ballot(j < best_d_j && labels[j] != la));
    }
    result = rank;
  }
  if (lane == 0) neg[pr] = result;
}

// The negatives of the reference's 'all_semi_hard' branch (loss/triplet_loss.py:158-183): per (anchor, positive) pair K distinct rows among
// the first L = max(K, #{j negative : d(a,p) + margin - d(a,j) > 0}) rows of the anchor's negatives list (ascending row order) — the
// reference draws K distinct ENUMERATE indices of its semi-hard list, topped up with the closest negatives when that list is short.
// One wave per pair; u[pair][K] uniforms in [0, 1) stand in for its random.sample: draw t takes the floor(u_t (L - t))-th position not
// taken yet (a uniform K-subset of the L positions, in draw order).  A pair whose anchor has fewer than K negatives raises *status
// (the reference's topk fails there) and gets -1.
constexpr int TSK_MAX = 8;
__global__ void triplet_select_k_kernel(const float* __restrict__ Dm, const int64_t* __restrict__ labels, int n, const int32_t* __restrict__ anc,
                                        const int32_t* __restrict__ pos, int P, float margin, int K, const float* __restrict__ u,
                                        int32_t* __restrict__ neg, int32_t* __restrict__ status) {
  const int pr = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (pr >= P) return;
  const int lane = threadIdx.x & 63;
  const int a = anc[pr];
  const int64_t la = labels[a];
  const float* row = Dm + (int64_t)a * n;
  const float thr = row[pos[pr]] + margin;
  int cnt_neg = 0, cnt_c = 0;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    const bool isneg = j < n && labels[j] != la;
    const bool c = isneg && (thr - row[j] > 0.f);
    cnt_neg += __popcll(__ballot(isneg));
    cnt_c += __popcll(__ballot(c));
  }
  if (cnt_neg < K) {
    if (lane == 0) { atomicMax(status, 1); for (int t = 0; t < K; ++t) neg[(int64_t)pr * K + t] = -1; }
    return;
  }
  const int L = cnt_c > K ? cnt_c : K;
  // K distinct positions in [0, L): wave-uniform arithmetic (every lane computes the same)
  int chosen[TSK_MAX], sorted[TSK_MAX];
  for (int t = 0; t < K; ++t) {
    int rsel = (int)(u[(int64_t)pr * K + t] * (float)(L - t));
    if (rsel >= L - t) rsel = L - t - 1;
    for (int q = 0; q < t; ++q)                                  // skip the positions taken so far, ascending
      if (rsel >= sorted[q]) ++rsel;
    chosen[t] = rsel;
    int q = t;
    while (q > 0 && sorted[q - 1] > rsel) { sorted[q] = sorted[q - 1]; --q; }
    sorted[q] = rsel;
  }
  // position in the negatives list -> row
  int seen = 0;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    const unsigned long long m = __ballot(j < n && labels[j] != la);
    const int pc = __popcll(m);
    for (int t = 0; t < K; ++t) {
      const int rk = chosen[t] - seen;
      if (rk >= 0 && rk < pc) {
        unsigned long long mm = m;
        for (int k = 0; k < rk; ++k) mm &= mm - 1;
        if (lane == 0) neg[(int64_t)pr * K + t] = j0 + (__ffsll((long long)mm) - 1);
      }
    }
    seen += pc;
  }
}

static inline hipStream_t S_(void* s) { return (hipStream_t)s; }

extern "C" int slic_triplet_select_k(const float* dist, const int64_t* labels, int n, const int32_t* anchors, const int32_t* positives, int P,
                                     float margin, int K, const float* u, int32_t* negatives, int32_t* status, void* stream) {
  SLIC_REQUIRE(dist && labels && anchors && positives && negatives && u && status && n > 1 && P > 0 && K >= 1 && K <= TSK_MAX,
               "slic_triplet_select_k: bad args (1 <= K <= %d)", TSK_MAX);
  triplet_select_k_kernel<<<dim3((unsigned)slic_cdiv(P, 4)), dim3(256), 0, S_(stream)>>>(dist, labels, n, anchors, positives, P, margin, K, u,
                                                                                          negatives, status);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_triplet_select(const float* dist, const int64_t* labels, int n, const int32_t* anchors,
                                   const int32_t* positives, int P, float margin, int mode, const float* u,
                                   int32_t* negatives, void* stream) {
  SLIC_REQUIRE(dist && labels && anchors && positives && negatives && n > 1 && P > 0 && mode >= 0 && mode <= 2 &&
               (mode == 2 || u), "slic_triplet_select: bad args");
  triplet_select_kernel<<<dim3((unsigned)slic_cdiv(P, 4)), dim3(256), 0, S_(stream)>>>(dist, labels, n, anchors, nullptr, positives, P, margin, mode, u, negatives);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_triplet_select_cross(const float* dist, const int64_t* col_labels, int ncols, const int32_t* anchor_rows,
                                         const int64_t* anchor_labels, const int32_t* positive_cols, int P, float margin,
                                         int mode, const float* u, int32_t* negatives, void* stream) {
  SLIC_REQUIRE(dist && col_labels && anchor_rows && anchor_labels && positive_cols && negatives && ncols > 0 && P > 0 &&
               mode >= 0 && mode <= 3 && (mode >= 2 || u), "slic_triplet_select_cross: bad args");
  triplet_select_kernel<<<dim3((unsigned)slic_cdiv(P, 4)), dim3(256), 0, S_(stream)>>>(dist, col_labels, ncols, anchor_rows, anchor_labels,
                                                                                        positive_cols, P, margin, mode, u, negatives);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}


extern "C" int slic_margin_cos_fwd(const float* X, const float* Y, const float* Z, int n, int D, float margin,
                                   float* state, float* rowloss, float* loss, void* stream) {
  SLIC_REQUIRE(X && Y && Z && state && rowloss && loss && n > 0 && D > 0, "slic_margin_cos_fwd: bad args");
  margin_cos_fwd_kernel<<<dim3((unsigned)slic_cdiv(n, 4)), dim3(256), 0, S_(stream)>>>(X, Y, Z, n, D, margin, state, rowloss);
  SLIC_LAUNCH_CHECK();
  mean_serial<<<dim3(1), dim3(64), 0, S_(stream)>>>(rowloss, n, loss);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
extern "C" int slic_margin_cos_bwd(const float* X, const float* Y, const float* Z, const float* state, int n, int D,
                                   const float* gscale, float* dX, float* dY, float* dZ, void* stream) {
  SLIC_REQUIRE(X && Y && Z && state && dX && dY && dZ && n > 0 && D > 0, "slic_margin_cos_bwd: bad args");
  margin_cos_bwd_kernel<<<dim3((unsigned)slic_cdiv(n, 4)), dim3(256), 0, S_(stream)>>>(X, Y, Z, state, n, D, gscale, dX, dY, dZ);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}


// ---- the euclidean forms (LOSS.DIST_METRIC: 'euclidean'; loss/triplet_loss.py:68-70, 212-214, online_train.py:289-291, 317-319, 345-347)
// margin term on F.pairwise_distance (eps 1e-6 inside the norm): v = |x - y + eps| - |x - z + eps| + margin, loss = mean max(0, v).
// One wave per row; state = {d1, d2, -, -, -, active}.
__global__ void margin_euclid_fwd_kernel(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ Z, int n,
                                         int D, float margin, float* __restrict__ st /* [n][8] */, float* __restrict__ rowloss) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  const float *x = X + (int64_t)row * D, *y = Y + (int64_t)row * D, *z = Z + (int64_t)row * D;
  float a = 0.f, b = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float dy = x[k] - y[k] + 1e-6f, dz = x[k] - z[k] + 1e-6f;
    a = fmaf(dy, dy, a); b = fmaf(dz, dz, b);
  }
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
  if (lane == 0) {
    const float d1 = sqrtf(a), d2 = sqrtf(b), v = d1 - d2 + margin;
    float* s = st + (int64_t)row * 8;
    s[0] = d1; s[1] = d2; s[5] = v > 0.f ? 1.f : 0.f;
    rowloss[row] = fmaxf(v, 0.f);
  }
}
__global__ void margin_euclid_bwd_kernel(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ Z,
                                         const float* __restrict__ st, int n, int D, const float* __restrict__ gscale,
                                         float* __restrict__ dX, float* __restrict__ dY, float* __restrict__ dZ) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  const float* s = st + (int64_t)row * 8;
  const float g = s[5] * (gscale ? *gscale : 1.f) / (float)n;
  const float i1 = s[0] > 0.f ? g / s[0] : 0.f, i2 = s[1] > 0.f ? g / s[1] : 0.f;      // a zero norm has a zero gradient (torch)
  const float *x = X + (int64_t)row * D, *y = Y + (int64_t)row * D, *z = Z + (int64_t)row * D;
  for (int k = lane; k < D; k += 64) {
    const float uy = (x[k] - y[k] + 1e-6f) * i1, uz = (x[k] - z[k] + 1e-6f) * i2;
    dX[(int64_t)row * D + k] = uy - uz;
    dY[(int64_t)row * D + k] = -uy;
    dZ[(int64_t)row * D + k] = uz;
  }
}

// NT-Xent on euclidean "similarities" (loss/triplet_loss.py:97-116 with dist_metric 'euclidean'): s_ij = (1 - |e_i - e_j|) / T,
// the diagonal filled with 0 before the division, target (n / 2 + i) mod n, loss = mean_i (logsumexp_j s_ij - s_i,target).
// One workgroup per row i.  Kept for the backward: dist [n][n] and wm [n][n] = d loss_i / d dist_ij = -(softmax_ij - [j = target]) / T
// (0 on the diagonal: masked_fill_ cuts the gradient there).
__global__ __launch_bounds__(256) void ntxent_euclid_fwd_kernel(const float* __restrict__ E, int n, int D, float invT,
                                                                float* __restrict__ dist, float* __restrict__ wm,
                                                                float* __restrict__ rowloss) {
  extern __shared__ float sc[];                       // [n] the row's logits
  __shared__ float red[4];
  const int i = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* ei = E + (int64_t)i * D;
  for (int j = wave; j < n; j += 4) {
    const float* ej = E + (int64_t)j * D;
    float a = 0.f;
    for (int k = lane; k < D; k += 64) { const float d = ei[k] - ej[k]; a = fmaf(d, d, a); }
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) {
      const float d = sqrtf(a);
      dist[(int64_t)i * n + j] = d;
      sc[j] = j == i ? 0.f : (1.f - d) * invT;
    }
  }
  __syncthreads();
  float m = -INFINITY;
  for (int j = t; j < n; j += 256) m = fmaxf(m, sc[j]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float se = 0.f;
  for (int j = t; j < n; j += 256) se += expf(sc[j] - m);
  for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
  if (lane == 0) red[wave] = se;
  __syncthreads();
  se = red[0] + red[1] + red[2] + red[3];
  const float lse = m + logf(se);
  const int tgt = (n / 2 + i) % n;
  if (t == 0) rowloss[i] = lse - sc[tgt];
  for (int j = t; j < n; j += 256) {
    const float pj = expf(sc[j] - lse) - (j == tgt ? 1.f : 0.f);
    wm[(int64_t)i * n + j] = j == i ? 0.f : -pj * invT;
  }
}
// dE_i = g / n * sum_j (wm_ij + wm_ji) (e_i - e_j) / dist_ij   (pairs at distance 0 contribute nothing, as torch's norm backward)
__global__ __launch_bounds__(256) void ntxent_euclid_bwd_kernel(const float* __restrict__ E, const float* __restrict__ dist,
                                                                const float* __restrict__ wm, int n, int D,
                                                                const float* __restrict__ gscale, float* __restrict__ dE) {
  extern __shared__ float cf[];                       // [n] coefficients of row i
  const int i = blockIdx.x, t = threadIdx.x;
  const float g = (gscale ? *gscale : 1.f) / (float)n;
  for (int j = t; j < n; j += 256) {
    const float d = dist[(int64_t)i * n + j];
    cf[j] = d > 0.f ? g * (wm[(int64_t)i * n + j] + wm[(int64_t)j * n + i]) / d : 0.f;
  }
  __syncthreads();
  const float* ei = E + (int64_t)i * D;
  for (int k = t; k < D; k += 256) {
    const float x = ei[k];
    float a = 0.f;
    for (int j = 0; j < n; ++j) a = fmaf(cf[j], x - E[(int64_t)j * D + k], a);
    dE[(int64_t)i * D + k] = a;
  }
}

extern "C" int slic_margin_euclid_fwd(const float* X, const float* Y, const float* Z, int n, int D, float margin,
                                      float* state, float* rowloss, float* loss, void* stream) {
  SLIC_REQUIRE(X && Y && Z && state && rowloss && loss && n > 0 && D > 0, "slic_margin_euclid_fwd: bad args");
  margin_euclid_fwd_kernel<<<dim3((unsigned)slic_cdiv(n, 4)), dim3(256), 0, S_(stream)>>>(X, Y, Z, n, D, margin, state, rowloss);
  SLIC_LAUNCH_CHECK();
  mean_serial<<<dim3(1), dim3(64), 0, S_(stream)>>>(rowloss, n, loss);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
extern "C" int slic_margin_euclid_bwd(const float* X, const float* Y, const float* Z, const float* state, int n, int D,
                                      const float* gscale, float* dX, float* dY, float* dZ, void* stream) {
  SLIC_REQUIRE(X && Y && Z && state && dX && dY && dZ && n > 0 && D > 0, "slic_margin_euclid_bwd: bad args");
  margin_euclid_bwd_kernel<<<dim3((unsigned)slic_cdiv(n, 4)), dim3(256), 0, S_(stream)>>>(X, Y, Z, state, n, D, gscale, dX, dY, dZ);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
extern "C" int slic_ntxent_euclid_fwd(const float* E, int n, int D, float temperature, float* dist, float* wm, float* rowloss,
                                      float* loss, void* stream) {
  SLIC_REQUIRE(E && dist && wm && rowloss && loss && n >= 2 && n % 2 == 0 && n <= 8192 && D > 0 && temperature > 0.f,
               "slic_ntxent_euclid_fwd: bad args (n even, 2 <= n <= 8192)");
  ntxent_euclid_fwd_kernel<<<dim3((unsigned)n), dim3(256), (size_t)n * 4, S_(stream)>>>(E, n, D, 1.0f / temperature, dist, wm, rowloss);
  SLIC_LAUNCH_CHECK();
  mean_serial<<<dim3(1), dim3(64), 0, S_(stream)>>>(rowloss, n, loss);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
extern "C" int slic_ntxent_euclid_bwd(const float* E, const float* dist, const float* wm, int n, int D, const float* gscale,
                                      float* dE, void* stream) {
  SLIC_REQUIRE(E && dist && wm && dE && n >= 2 && n <= 8192 && D > 0, "slic_ntxent_euclid_bwd: bad args");
  ntxent_euclid_bwd_kernel<<<dim3((unsigned)n), dim3(256), (size_t)n * 4, S_(stream)>>>(E, dist, wm, n, D, gscale, dE);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_pair_distance(const float* X, const float* Y, int n, int D, int euclidean, float* out,
                                  void* stream) {
  SLIC_REQUIRE(X && Y && out && n > 0 && D > 0, "slic_pair_distance: bad args");
  pair_distance_kernel<<<dim3((unsigned)slic_cdiv(n, 4)), dim3(256), 0, S_(stream)>>>(X, Y, n, D, euclidean, out);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_pair_distance_bwd(const float* X, const float* Y, const float* g, int n, int D, int euclidean, float* dX,
                                      float* dY, void* stream) {
  SLIC_REQUIRE(X && Y && g && dX && dY && n > 0 && D > 0, "slic_pair_distance_bwd: bad args");
  pair_distance_bwd_kernel<<<dim3((unsigned)slic_cdiv(n, 4)), dim3(256), 0, S_(stream)>>>(X, Y, g, n, D, euclidean, dX, dY);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_pdist2(const float* X, int nx, const float* Y, int ny, int D, float eps, int euclidean, float* out,
                           void* stream) {
  SLIC_REQUIRE(X && Y && out && nx > 0 && ny > 0 && D > 0, "slic_pdist2: bad args");
  pdist2_kernel<<<dim3((unsigned)slic_cdiv((int64_t)nx * ny, 4)), dim3(256), 0, S_(stream)>>>(X, nx, Y, ny, D, eps, euclidean, out);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_infonce_rows_fwd(const float* X, const float* Y, int P, int NY, int D, float temperature, float* state,
                                     float* rowloss, float* loss, void* stream) {
  SLIC_REQUIRE(X && Y && state && rowloss && loss && P > 0 && NY >= 2 && NY <= SLIC_INFONCE_MAXY && D > 0 && temperature > 0.f,
               "slic_infonce_rows_fwd: bad args (2 <= NY <= %d)", SLIC_INFONCE_MAXY);
  infonce_rows_fwd_kernel<<<dim3((unsigned)slic_cdiv(P, 4)), dim3(256), 0, S_(stream)>>>(X, Y, P, NY, D, 1.0f / temperature, state, rowloss);
  SLIC_LAUNCH_CHECK();
  mean_serial<<<dim3(1), dim3(64), 0, S_(stream)>>>(rowloss, P, loss);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
extern "C" int slic_infonce_rows_bwd(const float* X, const float* Y, const float* state, int P, int NY, int D,
                                     float temperature, const float* gscale, float* dX, float* dY, void* stream) {
  SLIC_REQUIRE(X && Y && state && dX && dY && P > 0 && NY >= 2 && NY <= SLIC_INFONCE_MAXY && D > 0 && temperature > 0.f,
               "slic_infonce_rows_bwd: bad args");
  infonce_rows_bwd_kernel<<<dim3((unsigned)slic_cdiv(P, 4)), dim3(256), 0, S_(stream)>>>(X, Y, state, P, NY, D, 1.0f / temperature, gscale, dX, dY);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_pdist(const float* V, int n, int D, float eps, int euclidean, float* out, void* stream) {
  SLIC_REQUIRE(V && out && n > 0 && D > 0, "slic_pdist: bad args");
  pdist_kernel<<<dim3((unsigned)slic_cdiv((int64_t)n * n, 4)), dim3(256), 0, S_(stream)>>>(V, n, D, eps, euclidean, out);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}


extern "C" size_t slic_ntxent_workspace_bytes(int n, int D) {
  return slic_align_up((size_t)n * D * 4, 256) + 3 * slic_align_up((size_t)n * 4, 256);
}

// workspace keeps (e_hat, rnorm, lse, rowloss) for the backward call
extern "C" int slic_ntxent_fwd(const float* E, int n, int D, int lde, float temperature, float* loss,
                               void* workspace, void* stream) {
  SLIC_REQUIRE(E && loss && workspace && n >= 2 && D >= 2 && D % 2 == 0 && lde >= D && temperature > 0.f,
               "slic_ntxent_fwd: bad args (n >= 2, D even)");
  hipStream_t st = S_(stream);
  SlicCarver w(workspace);
  float* Eh = w.take<float>((size_t)n * D);
  float* rnorm = w.take<float>(n);
  float* lse = w.take<float>(n);
  float* rowloss = w.take<float>(n);
  ntxent_normalize<<<dim3((unsigned)slic_cdiv(n, 4)), dim3(256), 0, st>>>(E, n, D, lde, Eh, rnorm);
  SLIC_LAUNCH_CHECK();
  ntxent_fwd_kernel<<<dim3((unsigned)slic_cdiv(n, 32)), dim3(256), 0, st>>>(Eh, n, D, 1.0f / temperature, lse, rowloss);
  SLIC_LAUNCH_CHECK();
  mean_serial<<<dim3(1), dim3(64), 0, st>>>(rowloss, n, loss);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_ntxent_bwd(const void* workspace, int n, int D, float temperature, const float* gscale,
                               float* dE, int ldd, void* stream) {
  SLIC_REQUIRE(workspace && dE && n >= 2 && D >= 2 && ldd >= D, "slic_ntxent_bwd: bad args");
  SLIC_REQUIRE((size_t)(n + D + 4) * 4 <= 64 * 1024, "slic_ntxent_bwd: n + D too large for one workgroup's LDS");
  SlicCarver w((void*)workspace);
  float* Eh = w.take<float>((size_t)n * D);
  float* rnorm = w.take<float>(n);
  float* lse = w.take<float>(n);
  ntxent_bwd_kernel<<<dim3(n), dim3(256), (size_t)(n + D + 4) * 4, S_(stream)>>>(Eh, rnorm, lse, n, D, 1.0f / temperature, gscale, dE, ldd);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
