/*
 * slic_hip.h — C ABI of libslic_hip.so, the MI355X (gfx950) implementation of SLIC's
 * contrastive-training hot path (R3D-18 encoder fwd/bwd, InfoNCE losses, full-dataset k-means,
 * cosine top-k retrieval).
 *
 * The reference (rvl-lab-utoronto/video_similarity_search) is 100 % Python and has no FFI
 * layer; the library calls it makes on this path — torch.nn.Conv3d /
 * BatchNorm3d (cuDNN), F.cosine_similarity + F.cross_entropy, torch.index_select + bmm,
 * sklearn.cluster.KMeans, sklearn cosine_distances + argsort — are what these entry points
 * replace.  Each declaration cites the reference call site it stands in for
 * (paths relative to the reference repo root).
 *
 * Conventions (all entry points):
 *   - extern "C", return int: 0 = SLIC_OK, negative = error (slic_last_error() gives text,
 *     thread-local).  Nothing throws, nothing allocates or frees caller memory.
 *   - pointers are DEVICE pointers unless the name ends in _host; sizes are elements unless
 *     the name ends in _bytes; `stream` is a hipStream_t passed as void* (NULL = default
 *     stream); all work is asynchronous on that stream.
 *   - workspace: where a function needs scratch memory the caller provides it; the matching
 *     *_workspace_bytes() function gives the size.  Workspaces may be reused across calls on
 *     the same stream.
 *   - fp32 everywhere unless stated; labels/indices int32; row-major.
 */
#ifndef SLIC_HIP_H
#define SLIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLIC_OK 0
#define SLIC_EINVAL -1   /* bad argument / unsupported shape */
#define SLIC_EHIP -2     /* HIP runtime error (launch, memset, ...) */
#define SLIC_ENODEV -3   /* no gfx950 device visible */

/* library */
int slic_version(void);                 /* (major<<16)|(minor<<8)|patch */
const char* slic_last_error(void);      /* thread-local, never NULL */
int slic_device_check(void);            /* SLIC_OK iff the current device is gfx950 */

/* ------------------------------------------------------------------------------------------
 * k-means (clustering/cluster_masks.py:64-71 -> sklearn.cluster.KMeans, algorithm 'lloyd';
 * sklearn/cluster/_k_means_lloyd.pyx:168-218, _k_means_common.pyx:167-311, _kmeans.py:174-277,
 * 624-752).  Floating-point contract: oracle/kmeans_oracle.c header.
 * ---------------------------------------------------------------------------------------- */

/* cnorm[j] = k-ascending fmaf chain of c_j.c_j   (row_norms(centers, squared=True),
 * _k_means_lloyd.pyx:98).  C: [K, D] with row stride ldc. */
int slic_kmeans_cnorm(const float* C, int K, int D, int ldc, float* cnorm, void* stream);

/* E-step: labels[i] = argmin_j cnorm[j] - 2*x_i.c_j, strict '<', first index wins
 * (_update_chunk_dense, _k_means_lloyd.pyx:189-213).  X: [N, D] row stride ldx (D % 8 == 0,
 * ldx % 4 == 0, 16-byte aligned).  If labels_old != NULL, *n_changed (int32, device) is
 * incremented by the number of rows whose label differs from labels_old (caller zeroes it).
 * best_score (optional, [N]) receives the winning score. */
size_t slic_kmeans_assign_workspace_bytes(int64_t N, int K);
int slic_kmeans_assign(const float* X, int64_t N, int D, int ldx, const float* C, int K, int ldc,
                       const float* cnorm, int32_t* labels, const int32_t* labels_old,
                       int32_t* n_changed, float* best_score, void* workspace, void* stream);

/* M-step sums: sums[j,:] = sum of rows with label j, fp32, ASCENDING ROW ORDER (deterministic;
 * == sklearn's centers_new[label] += X[i] loop on one thread, _k_means_lloyd.pyx:215-218);
 * counts[j] = member count as float (weight_in_clusters). */
size_t slic_kmeans_accumulate_workspace_bytes(int64_t N, int K);
int slic_kmeans_accumulate(const float* X, int64_t N, int D, int ldx, const int32_t* labels,
                           int K, float* sums, float* counts, void* workspace, void* stream);

/* sums/counts <- sum over shards s = 0..n_shards-1 (in that order) of partial_sums + s*shard_stride
 * ([K*D] each) and partial_counts + s*shard_stride ([K] each): the fixed-order combine after an
 * all-gather of per-GPU partials (== sklearn's per-thread buffers reduced in
 * thread order, _k_means_lloyd.pyx:142-152). */
int slic_kmeans_combine_shards(const float* partial_sums, const float* partial_counts,
                               int64_t shard_stride, int n_shards, int K, int D, float* sums,
                               float* counts, void* stream);

/* dist[i] = k-ascending fmaf chain of (x_ik - C[labels[i],k])^2 (used by empty-cluster
 * relocation, _k_means_common.pyx:183, and inertia, _k_means_common.pyx:103-128). */
int slic_kmeans_dist_to_assigned(const float* X, int64_t N, int D, int ldx, const float* C,
                                 int ldc, const int32_t* labels, float* dist, void* stream);

/* inertia = sum_i dist[i] in double (256-row blocks summed in order, then blocks in order). */
size_t slic_sum_f32_to_f64_workspace_bytes(int64_t N);
int slic_sum_f32_to_f64(const float* v, int64_t N, double* out, void* workspace, void* stream);

/* _relocate_empty_clusters_dense (_k_means_common.pyx:167-211), in two steps so that a sharded
 * run can merge candidates across GPUs in between:
 *  select_far: the n_sel farthest rows by (dist descending, ties -> lower row); dist is clobbered.
 *              (sklearn: np.argpartition(distances, -n_empty)[:-n_empty-1:-1]; identical for
 *              n_empty == 1, introselect-order-dependent above that.)
 *  apply_relocation: for e = 0..n-1 in order: sums[old_ids[e]] -= xfar[e]; sums[new_ids[e]] = xfar[e];
 *              counts[new] = 1; counts[old] -= 1.   xfar: [n, D] rows (stride ldf). */
int slic_kmeans_select_far(float* dist, int64_t N, int n_sel, int32_t* far_idx, float* far_dist,
                           void* stream);
int slic_kmeans_apply_relocation(const float* xfar, int ldf, const int32_t* old_ids,
                                 const int32_t* new_ids, int n, int D, float* sums, float* counts,
                                 void* stream);

/* _average_centers + _center_shift (_k_means_common.pyx:274-311): C_new = sums * (1/counts)
 * (empty clusters copy the biggest one, in sklearn's in-place order), shift[j] =
 * ||C_new_j - C_old_j||.  status (device double[4]) = { sum_j shift[j]^2, number of clusters with
 * counts == 0, *n_changed (or -1 if NULL), 0 } — the one word the host loop reads per iteration
 * (_kmeans.py:717-731).  C_new must not alias sums or C_old. */
int slic_kmeans_finalize(const float* C_old, const float* sums, const float* counts, int K, int D,
                         float* C_new, float* shift /* [K] */, const int32_t* n_changed,
                         double* status, void* stream);

/* column sums / sums of squares in double (rows ascending in 1024-row segments, segments in
 * order) — X.mean(axis=0) and np.var(X, axis=0) of KMeans.fit / _tolerance
 * (_kmeans.py:1479-1481, 279-288). */
size_t slic_col_stats_workspace_bytes(int64_t N, int D);
int slic_col_stats(const float* X, int64_t N, int D, int ldx, double* col_sum, double* col_sumsq,
                   void* workspace, void* stream);

/* out[i,:] = X[i,:] - v  (X -= X_mean, _kmeans.py:1481) */
int slic_sub_rowvec(const float* X, int64_t N, int D, int ldx, const float* v, float* out,
                    int ldo, void* stream);

/* out[i,:] = X[i,:] / ||X[i,:]||_2   (preprocess_features_kmeans, cluster_masks.py:30-34;
 * no epsilon, like the reference) */
int slic_l2norm_rows(const float* X, int64_t N, int D, int ldx, float* out, int ldo, void* stream);

/* k-means++ helper (_kmeans.py:236-264): for T candidate rows cand[t] of X,
 * newdist[t,i] = min(closest[i], ||x_i - x_cand[t]||^2) and pot[t] = sum_i newdist[t,i] (double).
 * If T == 1 and closest == NULL: newdist[0,i] = ||x_i - x_cand||^2 (first centre). */
size_t slic_kmeanspp_step_workspace_bytes(int64_t N, int T);
int slic_kmeanspp_step(const float* X, int64_t N, int D, int ldx, const int32_t* cand, int T,
                       const float* closest, float* newdist, double* pot, void* workspace,
                       void* stream);
/* inclusive prefix sum of v (float) in double, and searchsorted(cumsum, vals[t], 'left')
 * clipped to N-1  (stable_cumsum + np.searchsorted, _kmeans.py:243-248). */
size_t slic_cumsum_search_workspace_bytes(int64_t N);
int slic_cumsum_search(const float* v, int64_t N, const double* vals, int T, int32_t* idx_out,
                       void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SLIC_HIP_H */
