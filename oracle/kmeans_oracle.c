/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * CPU restatement (plain C + OpenMP) of the k-means path the reference calls at
 *   /root/reference/clustering/cluster_masks.py:64-71   KMeans(n_clusters=k, n_init=10).fit(X)
 * whose arithmetic lives in the third-party dependency scikit-learn (pinned ==0.22.0 in
 * /root/reference/requirements.txt:5; restated here from the 1.7.2 sources installed in
 * the build container, algorithm='lloyd'):
 *   sklearn/cluster/_kmeans.py:624-752           _kmeans_single_lloyd   (loop + stopping rules)
 *   sklearn/cluster/_k_means_lloyd.pyx:168-218   _update_chunk_dense    (E-step + M-step sums)
 *   sklearn/cluster/_k_means_common.pyx:167-211  _relocate_empty_clusters_dense
 *   sklearn/cluster/_k_means_common.pyx:274-295  _average_centers
 *   sklearn/cluster/_k_means_common.pyx:298-311  _center_shift
 *   sklearn/cluster/_kmeans.py:279-288           _tolerance
 *
 * Parity status: the reference holds NO test that pins k-means results, so
 * this oracle is pinned by golden vectors generated from sklearn 1.7.2 in the build
 * container (tests/golden/make_goldens_kmeans.py -> tests/golden/kmeans_*.npz).
 *
 * Floating-point contract (what "bit-exact" means between this file and the HIP path):
 *   score(i,j) = cnorm[j] - 2*dot(x_i, c_j)            (sklearn: ||c||^2 - 2 x.c via sgemm)
 *   dot(.,.)   = k-ascending chain of single-rounded fmaf, starting from +0
 *                (== what v_mfma_f32_32x32x2_f32 computes along K; OpenBLAS' blocking differs
 *                 by ulps — label agreement with sklearn itself is checked on the goldens)
 *   cnorm[j]   = the same chain on (c_j, c_j)
 *   argmin     = strict '<', first index wins           (_k_means_lloyd.pyx:205-213)
 *   M-step     = per-cluster fp32 sums in ascending row order inside a shard, shards added
 *                in shard order (== sklearn with n_threads = n_shards, static schedule);
 *                mean = sum * (float)(1.0/(double)count) (_average_centers)
 *   shift_j    = sqrtf of the 4-unrolled squared distance of _euclidean_dense_dense
 *                (_k_means_common.pyx:20-45), no FMA contraction
 * Compile with -ffp-contract=off so nothing but the explicit fmaf() fuses.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define JB 64 /* centroid block for the vectorisable inner loop */

int slic_oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* cnorm[j] = fmaf chain of c_j . c_j, k ascending */
void slic_oracle_row_sqnorm_chain(const float* C, int K, int D, float* out) {
  for (int j = 0; j < K; ++j) {
    float acc = 0.0f;
    const float* c = C + (size_t)j * D;
    for (int k = 0; k < D; ++k) acc = fmaf(c[k], c[k], acc);
    out[j] = acc;
  }
}

/*
 * E-step.  labels[i] = argmin_j cnorm[j] - 2*dot(x_i,c_j).  best_score (optional) gets the
 * winning score, second_score (optional) the runner-up (for top-2 gap reports).
 * Ct is the centroid matrix transposed to [D][Kp] (Kp = K rounded up to JB, zero padded) so
 * the j loop vectorises while every (i,j) chain stays strictly k-ordered.
 */
void slic_oracle_assign(const float* X, int64_t N, int D, const float* C, int K,
                        int32_t* labels, float* best_score, float* second_score) {
  int Kp = (K + JB - 1) / JB * JB;
  float* Ct = (float*)calloc((size_t)D * Kp, sizeof(float));
  float* cn = (float*)malloc(sizeof(float) * Kp);
  for (int j = 0; j < K; ++j)
    for (int k = 0; k < D; ++k) Ct[(size_t)k * Kp + j] = C[(size_t)j * D + k];
  slic_oracle_row_sqnorm_chain(C, K, D, cn);
#pragma omp parallel
  {
    float* acc = (float*)malloc(sizeof(float) * Kp);
#pragma omp for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
      const float* x = X + (size_t)i * D;
      for (int jb = 0; jb < Kp; jb += JB) {
        float a[JB];
        for (int j = 0; j < JB; ++j) a[j] = 0.0f;
        for (int k = 0; k < D; ++k) {
          const float xk = x[k];
          const float* ct = Ct + (size_t)k * Kp + jb;
          for (int j = 0; j < JB; ++j) a[j] = fmaf(xk, ct[j], a[j]);
        }
        for (int j = 0; j < JB; ++j) acc[jb + j] = a[j];
      }
      float best = cn[0] - 2.0f * acc[0];
      float second = INFINITY;
      int32_t lab = 0;
      for (int j = 1; j < K; ++j) {
        float s = cn[j] - 2.0f * acc[j];
        if (s < best) { second = best; best = s; lab = j; }
        else if (s < second) second = s;
      }
      labels[i] = lab;
      if (best_score) best_score[i] = best;
      if (second_score) second_score[i] = second;
    }
    free(acc);
  }
  free(Ct);
  free(cn);
}

/* shard s of n_shards owns rows [s*per, min(N,(s+1)*per)), per = ceil(N/n_shards) */
static void shard_range(int64_t N, int n_shards, int s, int64_t* lo, int64_t* hi) {
  int64_t per = (N + n_shards - 1) / n_shards;
  *lo = (int64_t)s * per; if (*lo > N) *lo = N;
  *hi = *lo + per;        if (*hi > N) *hi = N;
}

/* M-step sums (fp32, ascending rows inside a shard).  n_shards > 0: shards added in order in fp32 (an all-gather + ordered
 * add; == sklearn with n_threads = n_shards).  n_shards < 0: |n_shards| shards, their fp32 partials added in DOUBLE and the total
 * rounded to fp32 once — the arithmetic of an fp64 all-reduce of the partials, whose result does not depend on the reduction
 * order as long as the partials of one element lie within 2^29 of each other (the sum is then exact). */
void slic_oracle_accumulate(const float* X, int64_t N, int D, const int32_t* labels, int K,
                            int n_shards, float* sums, float* counts) {
  const int f64 = n_shards < 0;
  if (f64) n_shards = -n_shards;
  memset(sums, 0, sizeof(float) * (size_t)K * D);
  memset(counts, 0, sizeof(float) * K);
  float* ps = (float*)malloc(sizeof(float) * (size_t)K * D);
  float* pc = (float*)malloc(sizeof(float) * K);
  double* ds = f64 ? (double*)calloc((size_t)K * D, sizeof(double)) : 0;
  double* dc = f64 ? (double*)calloc(K, sizeof(double)) : 0;
  for (int s = 0; s < n_shards; ++s) {
    int64_t lo, hi; shard_range(N, n_shards, s, &lo, &hi);
    memset(ps, 0, sizeof(float) * (size_t)K * D);
    memset(pc, 0, sizeof(float) * K);
    for (int64_t i = lo; i < hi; ++i) {
      float* d = ps + (size_t)labels[i] * D;
      const float* x = X + (size_t)i * D;
      pc[labels[i]] += 1.0f;
      for (int k = 0; k < D; ++k) d[k] += x[k];
    }
    if (f64) {
      for (int j = 0; j < K; ++j) dc[j] += (double)pc[j];
      for (size_t e = 0; e < (size_t)K * D; ++e) ds[e] += (double)ps[e];
    } else {
      for (int j = 0; j < K; ++j) counts[j] += pc[j];
      for (size_t e = 0; e < (size_t)K * D; ++e) sums[e] += ps[e];
    }
  }
  if (f64) {
    for (int j = 0; j < K; ++j) counts[j] = (float)dc[j];
    for (size_t e = 0; e < (size_t)K * D; ++e) sums[e] = (float)ds[e];
    free(ds); free(dc);
  }
  free(ps); free(pc);
}

/* squared distance of every row to the centre it is assigned to: k-ascending fmaf chain of
 * (x-c)^2.  (sklearn: ((X - centers_old[labels])**2).sum(axis=1), numpy pairwise sum.) */
void slic_oracle_dist_to_assigned(const float* X, int64_t N, int D, const float* C,
                                  const int32_t* labels, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < N; ++i) {
    const float* x = X + (size_t)i * D;
    const float* c = C + (size_t)labels[i] * D;
    float acc = 0.0f;
    for (int k = 0; k < D; ++k) { float d = x[k] - c[k]; acc = fmaf(d, d, acc); }
    out[i] = acc;
  }
}

/*
 * _relocate_empty_clusters_dense.  Far points are taken in order of decreasing distance,
 * ties broken by the lower row index.  For one empty cluster this is exactly sklearn; for
 * several, sklearn's order is whatever numpy's introselect leaves in
 * argpartition(d,-n)[:-n-1:-1] (unspecified), so agreement is up to a permutation of the
 * relocated ids.  Returns the number of empty clusters found.
 */
int slic_oracle_relocate_empty(const float* X, int64_t N, int D, const float* C_old,
                               const int32_t* labels, int K, float* sums, float* counts) {
  int n_empty = 0;
  for (int j = 0; j < K; ++j) n_empty += (counts[j] == 0.0f);
  if (!n_empty) return 0;
  float* dist = (float*)malloc(sizeof(float) * N);
  slic_oracle_dist_to_assigned(X, N, D, C_old, labels, dist);
  float mx = 0.0f;
  for (int64_t i = 0; i < N; ++i) if (dist[i] > mx) mx = dist[i];
  if (mx == 0.0f) { free(dist); return n_empty; }
  int* empty = (int*)malloc(sizeof(int) * n_empty);   /* fixed up front, ascending (np.where) */
  for (int j = 0, e = 0; j < K; ++j) if (counts[j] == 0.0f) empty[e++] = j;
  for (int e = 0; e < n_empty; ++e) {
    int new_id = empty[e];
    int64_t far = 0; float fd = -1.0f;
    for (int64_t i = 0; i < N; ++i) if (dist[i] > fd) { fd = dist[i]; far = i; }
    dist[far] = -2.0f;
    int old_id = labels[far];
    const float* x = X + (size_t)far * D;
    for (int k = 0; k < D; ++k) {
      sums[(size_t)old_id * D + k] -= x[k];
      sums[(size_t)new_id * D + k] = x[k];
    }
    counts[new_id] = 1.0f;
    counts[old_id] -= 1.0f;
  }
  free(empty);
  free(dist);
  return n_empty;
}

/*
 * _average_centers + _center_shift, in sklearn's in-place j-ascending order (an empty cluster
 * copies centres[argmax_weight] as it stands at that moment: already averaged when
 * argmax_weight < j, still the raw sum otherwise).  `sums` becomes the new centres.
 * shift[j] = sqrtf(4-unrolled squared distance).  Returns sum_j shift[j]^2 in double.
 */
double slic_oracle_finalize(const float* C_old, float* sums, const float* counts, int K, int D,
                            float* shift) {
  int amax = 0;
  for (int j = 1; j < K; ++j) if (counts[j] > counts[amax]) amax = j;
  for (int j = 0; j < K; ++j) {
    float* c = sums + (size_t)j * D;
    if (counts[j] > 0.0f) {
      float alpha = (float)(1.0 / (double)counts[j]);
      for (int k = 0; k < D; ++k) c[k] *= alpha;
    } else {
      const float* src = sums + (size_t)amax * D;
      for (int k = 0; k < D; ++k) c[k] = src[k];
    }
  }
  double tot = 0.0;
  for (int j = 0; j < K; ++j) {
    const float* a = sums + (size_t)j * D;
    const float* b = C_old + (size_t)j * D;
    float r = 0.0f;
    int n4 = D / 4, k = 0;
    for (int q = 0; q < n4; ++q, k += 4) {
      float t = (a[k] - b[k]) * (a[k] - b[k]) + (a[k + 1] - b[k + 1]) * (a[k + 1] - b[k + 1]) +
                (a[k + 2] - b[k + 2]) * (a[k + 2] - b[k + 2]) +
                (a[k + 3] - b[k + 3]) * (a[k + 3] - b[k + 3]);
      r += t;
    }
    for (; k < D; ++k) r += (a[k] - b[k]) * (a[k] - b[k]);
    float sh = sqrtf(r);
    if (shift) shift[j] = sh;
    tot += (double)sh * (double)sh;
  }
  return tot;
}

/* inertia = sum_i ||x_i - c_label||^2, each row as in slic_oracle_dist_to_assigned, rows
 * added in double in ascending order */
double slic_oracle_inertia(const float* X, int64_t N, int D, const float* C,
                           const int32_t* labels) {
  float* dist = (float*)malloc(sizeof(float) * N);
  slic_oracle_dist_to_assigned(X, N, D, C, labels, dist);
  double t = 0.0;
  for (int64_t i = 0; i < N; ++i) t += (double)dist[i];
  free(dist);
  return t;
}

/*
 * Column mean and the sklearn tolerance of _tolerance(X, tol) = mean(var(X,axis=0))*tol.
 * Sums are taken in double: rows ascending inside segments of 1024 rows, segments added in
 * order (the order the HIP path uses; sklearn/numpy use fp32 accumulators, the difference
 * is below 1 ulp of the fp32 mean for the sizes tested).  mean_out[k] = (float)(sum/N);
 * var is of the CENTRED data (x - mean_out) as KMeans.fit computes it after X -= X_mean.
 */
void slic_oracle_col_mean(const float* X, int64_t N, int D, float* mean_out) {
  for (int k = 0; k < D; ++k) {
    double tot = 0.0;
    for (int64_t s = 0; s < N; s += 1024) {
      double seg = 0.0;
      int64_t e = s + 1024 < N ? s + 1024 : N;
      for (int64_t i = s; i < e; ++i) seg += (double)X[(size_t)i * D + k];
      tot += seg;
    }
    mean_out[k] = (float)(tot / (double)N);
  }
}

double slic_oracle_mean_var(const float* Xc, int64_t N, int D) {
  double acc = 0.0;
  for (int k = 0; k < D; ++k) {
    double t1 = 0.0, t2 = 0.0;
    for (int64_t s = 0; s < N; s += 1024) {
      double s1 = 0.0, s2 = 0.0;
      int64_t e = s + 1024 < N ? s + 1024 : N;
      for (int64_t i = s; i < e; ++i) {
        double v = (double)Xc[(size_t)i * D + k];
        s1 += v; s2 += v * v;
      }
      t1 += s1; t2 += s2;
    }
    double m = t1 / (double)N;
    acc += t2 / (double)N - m * m;
  }
  return acc / (double)D;
}

/*
 * _kmeans_single_lloyd on already centred data.  centers: in = init, out = final centres.
 * trace (optional): [max_iter][N] labels of every executed iteration.
 * Returns n_iter; *strict = 1 on strict (labels-unchanged) convergence; *inertia_out set.
 * If max_iter_fixed != 0 the stopping tests are skipped and exactly max_iter iterations run
 * (throughput runs: CPU and GPU do identical work).
 */
int slic_oracle_lloyd(const float* X, int64_t N, int D, int K, float* centers, int max_iter,
                      double tol_abs, int n_shards, int max_iter_fixed, int32_t* labels,
                      int32_t* trace, int* strict, double* inertia_out, int* n_relocations) {
  float* cnew = (float*)malloc(sizeof(float) * (size_t)K * D);
  float* counts = (float*)malloc(sizeof(float) * K);
  float* shift = (float*)malloc(sizeof(float) * K);
  int32_t* old = (int32_t*)malloc(sizeof(int32_t) * N);
  for (int64_t i = 0; i < N; ++i) { labels[i] = -1; old[i] = -1; }
  int it = 0, st = 0, nrel = 0;
  for (it = 0; it < max_iter; ++it) {
    slic_oracle_assign(X, N, D, centers, K, labels, 0, 0);
    slic_oracle_accumulate(X, N, D, labels, K, n_shards, cnew, counts);
    nrel += slic_oracle_relocate_empty(X, N, D, centers, labels, K, cnew, counts) > 0;
    double shift_tot = slic_oracle_finalize(centers, cnew, counts, K, D, shift);
    memcpy(centers, cnew, sizeof(float) * (size_t)K * D);
    if (trace) memcpy(trace + (size_t)it * N, labels, sizeof(int32_t) * N);
    if (!max_iter_fixed) {
      if (memcmp(labels, old, sizeof(int32_t) * N) == 0) { st = 1; break; }
      if (shift_tot <= tol_abs) break;
    }
    memcpy(old, labels, sizeof(int32_t) * N);
  }
  int n_iter = it < max_iter ? it + 1 : max_iter;
  if (!st) slic_oracle_assign(X, N, D, centers, K, labels, 0, 0);
  if (strict) *strict = st;
  if (inertia_out) *inertia_out = slic_oracle_inertia(X, N, D, centers, labels);
  if (n_relocations) *n_relocations = nrel;
  free(cnew); free(counts); free(shift); free(old);
  return n_iter;
}
