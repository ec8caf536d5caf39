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
#define SLIC_ETIMEOUT -4 /* a bounded wait on a collective ran out (slic_comm_create_timeout, slic_comm_wait): the communicator is aborted */

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

/* The same E-step on operands whose rows are already in the MFMA's LDS order — inside every group of eight k's,
 * [k0 k2 k4 k6 | k1 k3 k5 k7] (slic_kmeans_permute_k8; the centres can come permuted out of slic_kmeans_finalize) — so
 * both tiles go HBM -> LDS by DMA.  Same arithmetic, bit-identical labels; same workspace size. */
int slic_kmeans_permute_k8(const float* X, int64_t N, int D, int ldx, float* Xp, int ldxp, void* stream);
int slic_kmeans_assign_perm(const float* Xp, int64_t N, int D, int ldxp, const float* Cp, int K, int ldcp,
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
 * ||C_new_j - C_old_j||; cnorm_new (optional, [K]) = slic_kmeans_cnorm(C_new), so the next E-step needs no extra
 * launch.  status (device double[4]) = { sum_j shift[j]^2, number of clusters with
 * counts == 0, *n_changed (or -1 if NULL), 0 } — the one word the host loop reads per iteration
 * (_kmeans.py:717-731).  C_new must not alias sums or C_old. */
int slic_kmeans_finalize(const float* C_old, const float* sums, const float* counts, int K, int D,
                         float* C_new, float* shift /* [K] */, float* cnorm_new /* [K] or NULL */,
                         float* C_new_perm /* [K][D] in slic_kmeans_permute_k8 order, or NULL */,
                         int spherical /* 1: C_new rows are L2-normalised means (spherical k-means,
                                          clustering/cluster_masks.py:73-77 -> spherecluster) */,
                         const int32_t* n_changed, double* status, void* stream);

/* One whole single-GPU Lloyd iteration enqueued by one call: *n_changed = 0; slic_kmeans_assign_perm(Xp, Cp_old,
 * cnorm_old -> labels, n_changed vs labels_old); slic_kmeans_accumulate(X, labels -> sums, counts);
 * slic_kmeans_finalize(-> C_new, Cp_new, cnorm_new, shift, status).  X / Xp share ldx; the centre matrices are dense
 * [K][D].  (A sharded fit calls the three pieces itself, with the all-gather between the last two.) */
size_t slic_kmeans_lloyd_step_workspace_bytes(int64_t N, int K);
int slic_kmeans_lloyd_step(const float* X, const float* Xp, int64_t N, int D, int ldx, const float* C_old,
                           const float* Cp_old, const float* cnorm_old, int K, int32_t* labels,
                           const int32_t* labels_old, int32_t* n_changed, float* sums, float* counts,
                           float* C_new, float* Cp_new, float* cnorm_new, float* shift, int spherical,
                           double* status, void* workspace, void* stream);

/* The SHARDED Lloyd iteration (rows of X split over the ranks, centres replicated) as two calls around its ONE collective —
 * the GPU form of what the reference does on rank 0 while the other ranks wait in a barrier (online_train.py:625-662), with the
 * exchange SURVEY.md §8e row 2 specifies: one message [K*D sums | K counts | n_changed] per iteration.
 *   slic_kmeans_lloyd_local : *n_changed = 0; slic_kmeans_assign_perm on this rank's rows; slic_kmeans_accumulate into `payload`
 *       = [K*D sums | K counts | n_changed & 0xFFFFF | n_changed >> 20]  (K*D + K + 2 numbers; fp32, or — payload_f64 — the same
 *       fp32 results widened to double: the operand of an fp64 all-reduce, whose sum of <= 2^29-apart fp32 values is exact and
 *       therefore independent of RCCL's reduction order).
 *   (caller) all-gather of the fp32 payloads [n_parts][stride], or all-reduce(sum) of the fp64 payload (n_parts = 1).
 *   slic_kmeans_lloyd_global: sums / counts = the payloads added in part order (fp32), or the reduced fp64 payload rounded to
 *       fp32; then slic_kmeans_finalize on them (status[2] = the summed n_changed).  sums / counts are also written out
 *       (empty-cluster relocation reads them). */
size_t slic_kmeans_lloyd_local_workspace_bytes(int64_t N, int K);
int slic_kmeans_lloyd_local(const float* X, const float* Xp, int64_t N, int D, int ldx, const float* Cp_old,
                            const float* cnorm_old, int K, int32_t* labels, const int32_t* labels_old, void* payload,
                            int payload_f64, void* workspace, void* stream);
int slic_kmeans_lloyd_global(const void* parts, int parts_f64, int64_t stride, int n_parts, const float* C_old, int K, int D,
                             float* sums, float* counts, float* C_new, float* Cp_new, float* cnorm_new, float* shift,
                             int spherical, double* status, void* stream);

/* The collective of the sharded iteration behind the C ABI: an opaque RCCL communicator, created and destroyed explicitly (the
 * library's only global state besides the last-error string), and an in-place sum all-reduce on the caller's stream — a thin wrapper
 * over ncclAllReduce (RCCL over xGMI; bound at first use with dlopen, no link-time dependency).  One process per GPU: rank 0 calls
 * slic_comm_unique_id and hands the SLIC_COMM_ID_BYTES bytes to the other ranks (any channel: torch.distributed, a file, MPI), then
 * every rank calls slic_comm_create on its device.  Stands in for the NCCL group the reference sets up in
 * misc/distributed_helper.py:8-26 where the path needs its one exchange (online_train.py:625-662 -> SURVEY.md §8e row 2). */
#define SLIC_COMM_ID_BYTES 128
typedef struct slic_comm slic_comm;
int slic_comm_unique_id(void* id_out /* SLIC_COMM_ID_BYTES */);
int slic_comm_create(const void* id, int world, int rank, slic_comm** out);
/* The same with a deadline (milliseconds; 0 = none): the communicator is created non-blocking and polled; if the peers have not all
 * joined in time — a rank died before the rendezvous, a wrong world size — it is aborted and SLIC_ETIMEOUT comes back instead of a
 * hang.  The deadline also bounds the enqueue of every later slic_allreduce_* of this communicator. */
int slic_comm_create_timeout(const void* id, int world, int rank, int timeout_ms, slic_comm** out);
int slic_allreduce_f32(slic_comm* comm, float* buf, int64_t n, void* stream);
int slic_allreduce_f64(slic_comm* comm, double* buf, int64_t n, void* stream);
/* Bounded wait for the collectives already enqueued: returns when everything enqueued on `stream` so far has run (SLIC_OK), when RCCL
 * reports an asynchronous error (SLIC_EHIP), or when timeout_ms (0 = none) has passed (SLIC_ETIMEOUT: a peer never joined) — in the two
 * failure cases the communicator has been ABORTED (its stuck kernel is released, every later call on it fails) so that the rank can
 * exit non-zero instead of hanging its peers' job.  What torch.distributed's process-group timeout is to the reference's collectives
 * (misc/distributed_helper.py:30-64). */
int slic_comm_wait(slic_comm* comm, void* stream, int timeout_ms);
/* The same bounded wait on ONE event (a hipEvent_t the caller recorded behind the work it wants finished — the sharded Lloyd loop records
 * one behind every iteration's collective + status read-back and waits for iteration `it` while iteration `it + 1` is already enqueued:
 * slic_comm_wait would drain that one too). */
int slic_comm_wait_event(slic_comm* comm, void* event, int timeout_ms);
/* give up on a communicator at once (ncclCommAbort) and free the handle: never blocks (what a process-exit hook calls) */
int slic_comm_abort(slic_comm* comm);
/* orderly teardown: ncclCommFinalize, polled to completion under the communicator's deadline (10 s when it has none), then
 * ncclCommDestroy; a lost peer or an asynchronous error ends in ncclCommAbort (SLIC_ETIMEOUT / SLIC_EHIP).  The handle is freed either way. */
int slic_comm_destroy(slic_comm* comm);

/* The same exchange as a ONE-SHOT all-to-all over peer-mapped device memory (csrc/oneshot.hip): where the ring all-reduce above is 2 (W - 1)
 * dependent steps over xGMI's point-to-point links, every rank here WRITES its payload into an inbox of each peer at once (one link per
 * peer), raises a flag per 32 KB chunk, waits for its peers' flags and adds the W payloads it then holds in rank order — one kernel per
 * exchange, no RCCL.  Sums are fp64 (exact for the k-means payload: every rank ends with bit-identical numbers).  Replaces the rank-0
 * k-means + barrier of online_train.py:625-662 together with slic_kmeans_lloyd_local / _global; process model of
 * misc/distributed_helper.py:30-37 (one process per GPU; two processes may also share ONE GPU, which is how a one-GPU box tests it).
 *   slic_oneshot_create : allocates the rank's inbox for payloads of up to max_n doubles (uncached device memory) on the current device
 *                         and writes its IPC handle (SLIC_IPC_HANDLE_BYTES) to handle_out; timeout_ms bounds every wait of an exchange
 *                         (0 = none);
 *   (caller)            : gathers the W handles in rank order over any channel (torch.distributed, a file, MPI);
 *   slic_oneshot_connect: maps the peers' inboxes (hipIpcOpenMemHandle; needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this driver);
 *   slic_allreduce_oneshot_f64: in-place sum of n doubles (buf 16-byte aligned) across the ranks, asynchronous on `stream`; every
 *                         rank must issue the same sequence of exchanges with the same n;
 *   slic_oneshot_check  : once the stream (or an event behind the exchange) has completed — SLIC_OK, or SLIC_ETIMEOUT when a wait ran out:
 *                         a peer never pushed (lost rank).  The kernel itself never hangs: it gives up after timeout_ms, records the
 *                         exchange number in a host-mapped word and finishes; the payloads since are garbage and the caller must raise.
 *   slic_oneshot_info   : out[4] = {world, rank, memory kind (0 uncached, 1 fine-grained, 2 plain device memory), exchanges issued}. */
#define SLIC_IPC_HANDLE_BYTES 64
typedef struct slic_oneshot slic_oneshot;
int slic_oneshot_create(int world, int rank, int64_t max_n, int timeout_ms, slic_oneshot** out, void* handle_out /* SLIC_IPC_HANDLE_BYTES */);
int slic_oneshot_connect(slic_oneshot* c, const void* all_handles /* world x SLIC_IPC_HANDLE_BYTES, rank order */);
int slic_allreduce_oneshot_f64(slic_oneshot* c, double* buf, int64_t n, void* stream);
int slic_oneshot_check(slic_oneshot* c);
int slic_oneshot_info(const slic_oneshot* c, int* out /* [4] */);
int slic_oneshot_destroy(slic_oneshot* c);

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
/* k-means++ seeding (_kmeans_plusplus, _kmeans.py:174-277) as ONE enqueued sequence: centre 0 = row `first`; for
 * c = 1..K-1: T candidates = searchsorted(cumsum(closest), uniforms[(c-1)*T + t] * current_pot) (clipped), squared distances
 * of every row to them, min with closest, potentials in double, first-minimum candidate kept.  `uniforms` (device,
 * [(K-1)*T] doubles in [0, 1)) are the host RNG's draws — the draws do not depend on the data, so the host never waits.
 * idx_out (device, [K]) receives the chosen row indices. */
size_t slic_kmeanspp_run_workspace_bytes(int64_t N, int T);
int slic_kmeanspp_run(const float* X, int64_t N, int D, int ldx, int first, int K, int T, const double* uniforms,
                      int32_t* idx_out,
                      const float* Xp /* optional: slic_kmeans_permute_k8(X), same ldx */,
                      const float* xnorm /* optional: [N] squared row norms; with Xp the distances run on the matrix
                                            pipe as |x|^2 + |c|^2 - 2 x.c (sklearn's euclidean_distances form) */,
                      void* workspace, void* stream);
/* The R initialisations of KMeans(n_init = R) (the n_init loop of KMeans.fit, _kmeans.py:1500-1530, as
 * clustering/cluster_masks.py:70-71 calls it) seeded in lock-step: firsts[r] (HOST array) = run r's first row, uniforms (device,
 * [R][K-1][T] doubles: the host RNG's draws in the order the sequential loop makes them), idx_out (device, [R][K]).  One pass
 * over X per centre for all runs together; every run's arithmetic is that of slic_kmeanspp_run (bit-identical picks).  Xp = the
 * k-permuted copy (slic_kmeans_permute_k8), xnorm = its squared row norms; R * T <= 160. */
size_t slic_kmeanspp_run_batch_workspace_bytes(int64_t N, int T, int R);
int slic_kmeanspp_run_batch(const float* Xp, const float* xnorm, int64_t N, int D, int ldx, int R, const int32_t* firsts, int K, int T,
                            const double* uniforms, int32_t* idx_out, void* workspace, void* stream);

/* inclusive prefix sum of v (float) in double, and searchsorted(cumsum, vals[t], 'left')
 * clipped to N-1  (stable_cumsum + np.searchsorted, _kmeans.py:243-248). */
size_t slic_cumsum_search_workspace_bytes(int64_t N);
int slic_cumsum_search(const float* v, int64_t N, const double* vals, int T, int32_t* idx_out,
                       void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * 3-D convolution / linear layers as a table-driven gather-GEMM on fp32 MFMA
 * (models/resnet.py:11-25,126-131 nn.Conv3d forward + autograd dgrad/wgrad -> cuDNN in the
 * reference; models/resnet.py:182-184 nn.Linear).  Activations are NDHWC fp32
 * ([B, T, H, W, C], C % 4 == 0), weights packed K-contiguous.  See csrc/conv.hip.
 * ---------------------------------------------------------------------------------------- */
typedef struct SlicConvArgs {
  const float* src;        /* gathered operand: [B, Ts, Hs, Ws, Cs] */
  const float* wgt;        /* packed weights [N][ldw], k = table column (conv_gemm only) */
  float* dst;              /* output (conv_gemm only) */
  const int32_t* tab;      /* [nchunks][4] per 16-byte K-chunk: {src element delta,
                              tap mask (1<<(oa+3)) | (1<<(7+ob+3)) | (1<<(14+oc+3)) for tap offsets |o| <= 3, or -1 = all-zero chunk,
                              weight column of the chunk,
                              packed tap offsets (oa+128) | (ob+128)<<8 | (oc+128)<<16} */
  const int32_t* tap_tab;  /* optional [ntaps][4] per-TAP records {src element delta, tap mask, weight column base, 0} for the
                              LDS-DMA variants (needs Cs % 32 == 0, K = ntaps * Cs exactly); NULL otherwise */
  const float* bias;       /* [N] or NULL */
  const float* scale;      /* [N] or NULL: v = v*scale + shift (eval-mode BatchNorm) */
  const float* shift;      /* [N] or NULL */
  const float* addend;     /* same addressing as dst, or NULL: v += addend (residual / grad accumulate) */
  float* stat_partial;     /* [ceil(M/tile_m)][2][N] per-workgroup (sum, sum (v - mean_wg)^2) of v = acc+bias, or NULL */
  int64_t M;               /* rows = B * Ga * Gb * Gc */
  uint32_t src_bytes;      /* size of the src tensor in bytes (< 4 GiB: loads are range-checked buffer loads) */
  uint32_t wgt_bytes;      /* size of the packed weight matrix in bytes */
  int N;                   /* output channels */
  int nchunks;             /* K / 4, multiple of 8 */
  int Cs, Ts, Hs, Ws;      /* source dims */
  int Ga, Gb, Gc;          /* output grid per batch item */
  int sa, sb, sc;          /* source coordinate of grid point g = g*s (+ tap offset) */
  int ldw, ldo;            /* weight row stride, dst row stride (elements) */
  int dst_strided;         /* 0: dst row = m; 1: dst row = ((b*Da + ga*da+ea)*Db + gb*db+eb)*Dc + gc*dc+ec */
  int Da, Db, Dc, da, db, dc, ea, eb, ec;
  int relu;                /* clamp at 0 last */
  const float* mask_src;   /* optional, same addressing as dst: v = mask_src > 0 ? v : 0 after the addend (ReLU backward of the
                              layer that consumes this gradient) */
  const float* bwd_z;      /* optional, same addressing as dst: with bwd_mean / bwd_invstd ([N]) and bwd_partial, the epilogue
                              also emits the BatchNorm-backward partial sums of the stored gradient v: */
  const float* bwd_mean;
  const float* bwd_invstd;
  float* bwd_partial;      /* [ceil(M/tile_m)][2][N]: per workgroup (sum v, sum v * (bwd_z - mean) * invstd); NULL = off */
  const uint32_t* row_tab; /* optional [M][2] per-row records {byte offset of the row's source origin, 21-bit in-bounds mask}
                              written by slic_conv_row_table; slic_conv_wgrad's LDS-DMA kernel then does no per-row
                              coordinate arithmetic.  NULL: the kernel decodes rows itself */
  int k_run_len;           /* 0: K = taps x Cs, one pixel per tap (Cs % 4 == 0).  > 0: "W-run" operand for few-channel inputs
                              (the RGB stem): src is [B, Ts, Hs, Ws, Cs] with ANY Cs (3) and Ws already zero-padded along W; K is
                              cut into runs of k_run_len floats (a multiple of 4) = k_run_px consecutive pixels x Cs channels of one
                              (kt, kh) tap row, padded with zero-weight floats: 7 x 3 = 21 -> 24 instead of 7 x 4 = 28 */
  int k_run_px;            /* real pixels (kw taps) per run; run r covers taps r * k_run_px ... (slic_conv_wgrad unpacks with it) */
} SlicConvArgs;

/* rows per workgroup of the tile slic_conv_gemm runs for (args, variant) — sizes stat_partial / bwd_partial.  Variants:
 * 0 = register-staged 64 x 64 tiles (any source channel count: per-chunk table `tab`); 20 / 22 = LDS-DMA ring, 64 x 64 / 128 x 64 tiles
 * (per-tap table `tap_tab`, Cs % 32 == 0); 30 = Winograd F(4, 3) along W; 31 = Winograd F(4, 3) x F(2, 3) over (W, H) (see
 * slic_pack_weight_wino / slic_pack_weight_wino2). */
int slic_conv_tile_m(const SlicConvArgs* args, int variant);
/* dst = epilogue(gather(src) x wgt^T): forward conv, data gradient, linear. */
int slic_conv_gemm(const SlicConvArgs* args, int variant, void* stream);
/* n (<= 8) independent GEMMs with the same N in ONE launch (blockIdx.z picks the GEMM): the parity classes of a stride-2
 * data gradient, whose K loops are too short (1-8 taps) to fill the chip one launch at a time.  Variants 20 and 22. */
int slic_conv_gemm_multi(const SlicConvArgs* args, int n, int variant, void* stream);
/* Tail-split launch of the same GEMM (variants 20 and 22): row blocks [0, nfull_rb) run whole; the remaining ones — what
 * would be a last, partly filled round of the chip's residency slots — are cut `splits` ways along K into the same grid and
 * finished (pieces added in split order, then the ordinary epilogue) by a second pass over those rows only.  nfull_rb = 0 is plain
 * split-K (raw accumulators to workspace[splits][M][N]); splits <= 1 forwards to slic_conv_gemm.  Deterministic.  Stands in for cuDNN's algorithm choice on
 * the few-tile layers of models/resnet.py:126-131 (layer3 / layer4 at small M).
 * Variant 31 (two-dimensional Winograd): nfull_rb counts 64-TILE blocks; the blocks behind them cut their K loop — the 3 Cs / 8 double
 * stages in (channel group, kt) order — into `splits` even pieces (splits must divide 3 Cs / 16, 2 .. 16: SLIC_EINVAL otherwise, and a
 * workspace size of 0) and a finish pass adds the pieces in order and runs the epilogue; splits <= 1 is the plain launch. */
size_t slic_conv_gemm_tailsplit_workspace_bytes(const SlicConvArgs* args, int variant, int nfull_rb, int splits);
int slic_conv_gemm_tailsplit(const SlicConvArgs* args, int variant, int nfull_rb, int splits, void* workspace, void* stream);

/* dW[N][C][ntaps] (reference layout) = sum_m gather(src)[m, tap*Cs + c] * dy[m, n]; `splits` slices of
 * m reduced in fixed order.  args->wgt/dst unused. */
size_t slic_conv_wgrad_workspace_bytes(const SlicConvArgs* args, int splits);
int slic_conv_wgrad(const SlicConvArgs* args, const float* dy, int ldy, int splits, int C, int ntaps,
                    float* dW, void* workspace, void* stream);
/* row_tab[m] = {(((b*Ts + ga*sa)*Hs + gb*sb)*Ws + gc*sc)*Cs*4, bit (7*dim + o + 3) set iff coordinate + o is inside the
 * source for o in -3..3} for the M rows of args' geometry (args->src etc. unused): 8 bytes per row. */
int slic_conv_row_table(const SlicConvArgs* args, uint32_t* row_tab, void* stream);
/* Wp[n][tap*Cs + c] = W[n][c][tap] for c < C — forward operand.  The padding (channels C..Cs, columns ntaps*Cs..Kp) is NOT
 * written: the caller zeroes the operand once when it allocates it (same for Wd below: rows C..Cs, columns ntaps*N..Kd). */
int slic_pack_weight_fwd(const float* W, int N, int C, int ntaps, int Cs, int Kp, float* Wp, void* stream);
/* W-run operand (SlicConvArgs.k_run_len): Wp[n][run * run_len + px * C + c] = W[n][c][run * run_px + px], zero elsewhere */
int slic_pack_weight_fwd_runs(const float* W, int N, int C, int ntaps, int run_len, int run_px, int Kp, float* Wp, void* stream);
/* Wd[c][tap*N + n] = W[n][c][tap] (Cs rows, Kd columns) — data-gradient operand */
int slic_pack_weight_dgrad(const float* W, int N, int C, int ntaps, int Cs, int Kd, float* Wd, void* stream);
/* Operand of slic_conv_gemm variant 30 — Winograd F(4, 3) along W for the 3 x 3 x 3 / stride 1 / pad 1 layers (what cuDNN's
 * algorithm search may pick behind nn.Conv3d, models/resnet.py:11-17, online_train.py:444): U_p = sum_kw G[p][kw] w[..][kw] for the
 * six points p, laid out in the kernel's LDS stage order
 *   U[(((tap9 * C_/8 + cc) * N_/64 + nb) * 12 + p * 2 + h) * 64 + nl][j],  tap9 = kt * 3 + kh, n = 64 nb + nl, c = 8 cc + 4 h + j.
 * dgrad = 0: forward operand (N_ = N outputs, C_ = C reduction channels); dgrad = 1: data-gradient operand (N_ = C, C_ = N, taps
 * flipped).  9 * C * N * 6 floats.  Needs N_ % 64 == 0 and C_ % 8 == 0.  Variant 30 itself: same SlicConvArgs as the other
 * variants (stride-1 same-size geometry, no bias), wgt / wgt_bytes = this operand; Ws % 4 == 0, or — the last tile of a row ragged —
 * a padded width 4 ceil(Ws / 4) that divides 128; slab rows of slic_conv_tile_m(args, 30) GEMM rows (128, or 128 / Wp * Ws). */
int slic_pack_weight_wino(const float* W, int N, int C, int dgrad, float* U, void* stream);
/* Variant 31 of slic_conv_gemm: Winograd in TWO dimensions, F(4, 3) along W x F(2, 3) along H — 24 multiplies per (kt, c, n) and tile of
 * 2 x 4 outputs where the direct form has 72 and variant 30 has 36, all exact-fp32 MFMA.  Same SlicConvArgs as variant 30 (stride-1
 * same-size 3 x 3 x 3 geometry, no bias, Cs = 64 x a power of two, N % 64 == 0); wgt / wgt_bytes = the operand written here:
 * U2[kt][C/4][N/64][Hpoint 4][Wpoint 6][column half 2][channel pair 2][n 32][2 ch] = Gh w Gw^T, 3 * 24 * C * N floats (dgrad = 1: flipped / transposed for the data
 * gradient).  Stands in for the same cuDNN calls as variant 30 (models/resnet.py:11-17, 41-57).  Blocks of 64 tiles must all hold the
 * same number of real outputs — slic_conv_tile_m(args, 31) returns it (the slab rows' size) or 0 when the geometry does not allow
 * it: H even and W % 4 == 0, or H even and ceil(W / 4) | 64, or ceil(H / 2) * ceil(W / 4) | 64. */
int slic_pack_weight_wino2(const float* W, int N, int C, int dgrad, float* U, void* stream);
/* Weight gradient by the transposed TWO-dimensional algorithm, dW = Gh^T [sum over tiles of (Bh^T x Bw) . (Ah dY Aw^T)] Gw: 24
 * multiplies per (kt, c, n) and tile of 2 x 4 outputs (slic_conv_wgrad_wino: 36, the direct form: 72); replaces slic_conv_wgrad_wino
 * where variant 31 runs the forward (any H, W: ragged tiles are masked).  args as slic_conv_wgrad_wino; tile_tab: (M / (Hs Ws)) *
 * ceil(Hs / 2) * ceil(Ws / 4) records of 8 bytes written once per geometry by slic_conv_wino2_tile_table; `splits` slices of the
 * tiles, reduced in a fixed order (deterministic: groups of consecutive slices, then the groups); workspace: splits x 3 x 12 x Cs x N floats (the
 * W-points are taken back through Gw before they leave the kernel).  One workgroup per (kt, 64 x 64 block,
 * slice) holding ALL FOUR H-points: 3 x Cs / 64 x N / 64 x splits workgroups of 512 threads, one per CU.  Limits (SLIC_EINVAL beyond them;
 * models/conv_plan.py keeps such shapes on slic_conv_wgrad_wino / slic_conv_wgrad): M < 2^24 output positions, x and dy below 4 GiB - 256 B.
 * Stands in for the same autograd weight gradient of nn.Conv3d (models/resnet.py:11-17) as slic_conv_wgrad. */
size_t slic_conv_wgrad_wino2_workspace_bytes(const SlicConvArgs* args, int splits);
int slic_conv_wino2_tile_table(const SlicConvArgs* args, uint32_t* tile_tab, void* stream);
int slic_conv_wgrad_wino2(const SlicConvArgs* args, const float* dy, int splits, const uint32_t* tile_tab, float* dW,
                          void* workspace, void* stream);
/* Weight gradient of the same layers by the transposed F(4, 3) algorithm, dW[kw] = sum over W-tiles of G^T[(B^T x) . (A dy)]
 * (six multiplies per (kt, kh, c, n) and tile of four outputs instead of twelve); replaces slic_conv_wgrad where variant 30 runs the
 * forward.  args: the forward geometry (src = x, 3 x 3 x 3 / stride 1 / pad 1, Cs % 64 == 0, N % 64 == 0, any Ws: the last tile of a
 * row may be ragged); dy dense [M][N]; tile_tab: (M / Ws) * ceil(Ws / 4) records of 8 bytes written once per geometry by
 * slic_conv_wino_tile_table; `splits` slices of the tiles,
 * reduced in slice order (deterministic); dW in the reference layout [N][C][3][3][3].  workspace: splits x 9 x 6 x Cs x N floats. */
size_t slic_conv_wgrad_wino_workspace_bytes(const SlicConvArgs* args, int splits);
int slic_conv_wino_tile_table(const SlicConvArgs* args, uint32_t* tile_tab, void* stream);
int slic_conv_wgrad_wino(const SlicConvArgs* args, const float* dy, int splits, const uint32_t* tile_tab, float* dW,
                         void* workspace, void* stream);
/* [B, C, S] -> [B, S, Cp] with channels zero-padded to Cp (clip NCDHW -> NDHWC4, datasets/dataset_utils.py:104) */
int slic_ncdhw_to_ndhwc(const float* x, int B, int C, int64_t S, int Cp, float* y, void* stream);
/* [B, C, R, W] -> [B, R, Wp, C] (R = T*H rows): column w lands at w + pad_left, the other columns are zero (Wp >= W + pad_left):
 * the W-run stem operand */
int slic_ncdhw_to_ndhwc_wpad(const float* x, int B, int C, int64_t R, int W, int pad_left, int Wp, float* y, void* stream);

/* SyncBatchNorm (online_train.py:466-468 -> torch.nn.SyncBatchNorm.convert_sync_batchnorm): the rank-local halves of
 * slic_bn_finalize / slic_bn_bwd*, with the collective left to the caller (torch.distributed over RCCL).
 *   forward : slic_bn_merge_stats -> stats[0..C) = sum, stats[C..2C) = M2 (doubles); the caller writes its sample count to
 *             stats[2C], all-gathers the rows in rank order, and every rank runs slic_bn_finalize_sync on [W][2C+1]
 *             (Chan's merge in rank order; normalisation with the biased variance, running stats with the unbiased one over
 *             the global count).
 *   backward: slic_bn_bwd_sums -> sums[0..C) = sum g, sums[C..2C) = sum g*xhat (doubles, rank-local) + rank-local dgamma/dbeta,
 *             from a dgrad epilogue's slab (partial, R_partial) or, partial == NULL, from dy / out (ReLU mask) / z with the masked
 *             gradient written to g_out; the caller all-reduces `sums`, divides by the global count (ka, kb) and calls
 *             slic_bn_bwd_apply. */
int slic_bn_merge_stats(const float* partial, int R, int rows, int C, int64_t M, double* stats, void* workspace, void* stream);
int slic_bn_finalize_sync(const double* stats_all, int W, int C, float eps, float momentum, const float* gamma,
                          const float* beta, float* mean, float* invstd, float* scale, float* shift,
                          float* running_mean, float* running_var, void* stream);
size_t slic_bn_bwd_sums_workspace_bytes(int64_t M, int C, int R_partial);
int slic_bn_bwd_sums(const float* partial, int R_partial, const float* dy, const float* out, const float* z,
                     const float* mean, const float* invstd, int64_t M, int C, float* g_out, double* sums,
                     float* dgamma, float* dbeta, void* workspace, void* stream);
int slic_bn_bwd_apply(const float* g, const float* z, const float* mean, const float* invstd, const float* gamma,
                      const double* ka, const double* kb, int64_t M, int C, float* dz, void* stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm3d/1d + ReLU + residual + global average pool (models/resnet.py:34-57,132-133,173,183,
 * 294-299; torch defaults eps = 1e-5, momentum = 0.1).  Activations [M, C] row-major, C % 4 == 0.
 * ---------------------------------------------------------------------------------------- */
/* batch statistics from the conv epilogue's per-workgroup slab partial[R][2][C] = (sum, sum (x - mean_blk)^2)
 * over `rows` rows per workgroup (the last one ragged), merged in workgroup order in double (Chan's formula):
 * mean, invstd = 1/sqrt(biased var + eps), scale = gamma*invstd, shift = beta - mean*scale; running stats
 * (optional pair) get the momentum update with the unbiased variance. */
size_t slic_bn_finalize_workspace_bytes(int R, int C);
int slic_bn_finalize(const float* partial, int R, int rows, int C, int64_t M, float eps, float momentum,
                     const float* gamma, const float* beta, float* mean, float* invstd, float* scale,
                     float* shift, float* running_mean, float* running_var, void* workspace, void* stream);
/* eval mode: scale = gamma/sqrt(running_var+eps), shift = beta - running_mean*scale */
int slic_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, int C, float* scale, float* shift, void* stream);
/* y = relu?(z*scale + shift (+ res)) */
int slic_bn_apply(const float* z, const float* scale, const float* shift, const float* res, int relu,
                  int64_t M, int C, float* y, void* stream);
/* backward of y = relu?(BN(z) (+res)) in train mode: g = dy * (out > 0) when `out` (the saved post-ReLU
 * output) is given, else g = dy; g_out (optional) receives g (it is also the gradient of `res`);
 * dz = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)); dgamma = sum g*xhat; dbeta = sum g. */
size_t slic_bn_bwd_workspace_bytes(int64_t M, int C, int need_g_buffer);
int slic_bn_bwd_rows_per_partial(void);
int slic_bn_bwd(const float* dy, const float* out, const float* z, const float* mean, const float* invstd,
                const float* gamma, int64_t M, int C, float* g_out, float* dz, float* dgamma, float* dbeta,
                void* workspace, void* stream);
/* The second half of slic_bn_bwd when the producer of g (slic_conv_gemm with args->bwd_partial) already applied the
 * ReLU mask and emitted the R x [2][C] partial sums: merges them in slab order (double), writes dgamma / dbeta and
 * dz = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)). */
size_t slic_bn_bwd_fused_workspace_bytes(int R, int C);
int slic_bn_bwd_fused(const float* partial, int R, const float* g, const float* z, const float* mean,
                      const float* invstd, const float* gamma, int64_t M, int C, float* dz, float* dgamma,
                      float* dbeta, void* workspace, void* stream);
/* AdaptiveAvgPool3d(1): y[b,c] = mean_s x[b,s,c]; backward dx = dy / S */
int slic_avgpool_fwd(const float* x, int B, int S, int C, float* y, void* stream);
int slic_avgpool_bwd(const float* dy, int B, int S, int C, float* dx, void* stream);
/* the stem's nn.MaxPool3d(kernel_size=3, stride=2, padding=1) (models/resnet.py:123, 262-263; `no_max_pool=False`) on an NDHWC
 * tensor: y [B, To, Ho, Wo, C] with To = (T - 1) / 2 + 1 ...; arg (optional) = the winning input position per output element
 * (first maximum in (t, h, w) order), which the backward gathers from */
int slic_maxpool3d_fwd(const float* x, int B, int T, int H, int W, int C, float* y, int32_t* arg, void* stream);
int slic_maxpool3d_bwd(const float* dy, const int32_t* arg, int B, int T, int H, int W, int C, float* dx, void* stream);
/* shortcut_type 'A' (models/resnet.py:213-222): every stride-th position of x [B, T, H, W, C], channels zero-padded to C_out.
 * The reference concatenates `out.data`, so the branch carries no gradient: there is no backward entry */
int slic_shortcut_a(const float* x, int B, int T, int H, int W, int C, int stride, int C_out, float* y, void* stream);
/* out[c] = sum_m x[m,c] (rows ascending, double accumulator): nn.Linear bias gradient */
int slic_colsum(const float* x, int64_t M, int C, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * InfoNCE / NT-Xent (loss/triplet_loss.py:95-116 'noise_contrastive' + pdist :429-437):
 * loss = CE( (1 - (1 - cos(e_i, e_j))) with diag := 0, / T ;  target (n/2 + i) mod n ), mean over n rows.
 * fwd keeps (e_hat, 1/norm, row log-sum-exp) in `workspace` for bwd.  gscale: device scalar upstream
 * gradient (NULL = 1).  D even.
 * ---------------------------------------------------------------------------------------- */
size_t slic_ntxent_workspace_bytes(int n, int D);
int slic_ntxent_fwd(const float* E, int n, int D, int lde, float temperature, float* loss, void* workspace,
                    void* stream);
int slic_ntxent_bwd(const void* workspace, int n, int D, float temperature, const float* gscale, float* dE,
                    int ldd, void* stream);
/* rowwise distance of two [n, D] matrices: 1 - cos (per-norm clamp 1e-8) or ||x - y + 1e-6||_2
 * (models/triplet_net.py:29-33: F.cosine_similarity / F.pairwise_distance) */
int slic_pair_distance(const float* X, const float* Y, int n, int D, int euclidean, float* out, void* stream);
/* its backward: dX, dY [n, D] from g [n] = dL/d dist (the reference's distances are ordinary autograd nodes,
 * models/triplet_net.py:28-32) */
int slic_pair_distance_bwd(const float* X, const float* Y, const float* g, int n, int D, int euclidean, float* dX, float* dY,
                           void* stream);
/* LLC margin term of triplet_train_epoch (online_train.py:317-332): loss = mean_i max(0, (1 - cos(x_i, y_i)) -
 * (1 - cos(x_i, z_i)) + margin)  (MarginRankingLoss with target -1 on the two cosine distances).
 * state: [n, 8] floats kept for bwd; rowloss: [n] scratch.  bwd: gradients to all three inputs. */
int slic_margin_cos_fwd(const float* X, const float* Y, const float* Z, int n, int D, float margin, float* state,
                        float* rowloss, float* loss, void* stream);
int slic_margin_cos_bwd(const float* X, const float* Y, const float* Z, const float* state, int n, int D,
                        const float* gscale, float* dX, float* dY, float* dZ, void* stream);
/* the same margin term on euclidean distances, F.pairwise_distance(x, y) = ||x - y + 1e-6||_2 (LOSS.DIST_METRIC 'euclidean':
 * loss/triplet_loss.py:68-70, 212-214; online_train.py:289-291, 317-319, 345-347); state / rowloss as above */
int slic_margin_euclid_fwd(const float* X, const float* Y, const float* Z, int n, int D, float margin, float* state,
                           float* rowloss, float* loss, void* stream);
int slic_margin_euclid_bwd(const float* X, const float* Y, const float* Z, const float* state, int n, int D,
                           const float* gscale, float* dX, float* dY, float* dZ, void* stream);
/* 'noise_contrastive' with dist_metric 'euclidean' (loss/triplet_loss.py:97-116): logits (1 - ||e_i - e_j||) / T, diagonal 0,
 * target (n / 2 + i) mod n, mean cross-entropy.  dist, wm: [n, n] floats kept for the backward; rowloss: [n] scratch */
int slic_ntxent_euclid_fwd(const float* E, int n, int D, float temperature, float* dist, float* wm, float* rowloss, float* loss,
                           void* stream);
int slic_ntxent_euclid_bwd(const float* E, const float* dist, const float* wm, int n, int D, const float* gscale, float* dE,
                           void* stream);
/* negative selection of NegativeTripletSelector.get_one_one_triplets (loss/triplet_loss.py:311-360) for P (anchor,
 * positive) pairs over the [n, n] distance matrix `dist`: mode 0 random_negative, 1 random_semi_hard, 2 fixed_semi_hard;
 * u[P] uniform in [0,1) stands in for Python's random.choice (unused for mode 2); fallback = hardest easy negative,
 * returned as its position in the negatives list exactly like the reference.  labels: int64 [n]. */
int slic_triplet_select(const float* dist, const int64_t* labels, int n, const int32_t* anchors,
                        const int32_t* positives, int P, float margin, int mode, const float* u, int32_t* negatives,
                        void* stream);
/* the K (<= 8) negatives per pair of the reference's 'all_semi_hard' branch (loss/triplet_loss.py:158-183, K = 5 there): K distinct rows
 * among the first max(K, #semi-hard) rows of the anchor's negatives list (ascending rows); u[P][K] uniforms in [0, 1) stand in for
 * random.sample (draw t = the floor(u (L - t))-th position not taken yet).  negatives: int32 [P][K].  *status (int32, caller-zeroed) is
 * raised to 1 when an anchor has fewer than K negatives (the reference's topk fails there); such pairs get -1. */
int slic_triplet_select_k(const float* dist, const int64_t* labels, int n, const int32_t* anchors, const int32_t* positives, int P,
                          float margin, int K, const float* u, int32_t* negatives, int32_t* status, void* stream);
/* the same over a rectangular matrix dist[rows][ncols] (MemTripletLoss, loss/triplet_loss.py:9-81, 239-272: rows = the batch,
 * columns = the queue): col_labels int64 [ncols], the pair's anchor row / anchor label / positive COLUMN; mode 3 =
 * 'adapted_hard' (the reference's sampler returns nothing, so every pair takes the hardest-easy fallback). */
int slic_triplet_select_cross(const float* dist, const int64_t* col_labels, int ncols, const int32_t* anchor_rows,
                              const int64_t* anchor_labels, const int32_t* positive_cols, int P, float margin, int mode,
                              const float* u, int32_t* negatives, void* stream);
/* [n, n] distance matrix of the rows of V (loss/triplet_loss.py:429-437 pdist) */
int slic_pdist(const float* V, int n, int D, float eps, int euclidean, float* out, void* stream);
/* [nx, ny] distance matrix between the rows of X and Y (loss/triplet_loss.py:439-447 pdist_v2) */
int slic_pdist2(const float* X, int nx, const float* Y, int ny, int D, float eps, int euclidean, float* out, void* stream);
/* InfoNCE over gathered rows ('all_semi_hard', loss/triplet_loss.py:118-203): X [P, D] anchors, Y [P, NY, D] with row 0 the
 * positive and rows 1..NY-1 the picked negatives (NY <= 8): loss = mean_i -log(e^{cos(x,y0)/T} / sum_j e^{cos(x,yj)/T}).
 * state [P][18] floats is kept for the backward, which returns dX [P, D] and dY [P, NY, D] scaled by *gscale. */
int slic_infonce_rows_fwd(const float* X, const float* Y, int P, int NY, int D, float temperature, float* state,
                          float* rowloss, float* loss, void* stream);
int slic_infonce_rows_bwd(const float* X, const float* Y, const float* state, int P, int NY, int D, float temperature,
                          const float* gscale, float* dX, float* dY, void* stream);

/* ------------------------------------------------------------------------------------------
 * Cosine top-k retrieval (iic_retrieve_clips.py:275-314 topk_retrieval; evaluate.py:208-231,287-307):
 * sklearn cosine_distances + argsort/argpartition on the host become a fused similarity GEMM +
 * streaming top-k; the Nq x Ng matrix is never written.
 * ---------------------------------------------------------------------------------------- */
/* sklearn.preprocessing.normalize(X, 'l2'): out[i,:] = X[i,:] / ||X[i,:]|| (zero rows unchanged); out is dense [N, D] */
int slic_normalize_rows(const float* X, int64_t N, int D, int ldx, float* out, void* stream);
/* for every row of Qn the k nearest rows of Gn by cosine distance clip(1 - q.g, 0, 2), ascending (ties -> lower
 * gallery index).  Qn/Gn: normalised, dense [N, D], D % 8 == 0.  self_mask != 0 skips j == i
 * (np.fill_diagonal(distance_matrix, inf), evaluate.py:221-222).  k <= 88 (SLIC_EINVAL beyond: the per-query lists of the
 * streaming kernels live in LDS).  Two algorithms, identical results: galleries of >= 32768 rows with D <= 512 and k >= 16 go threshold ->
 * collect -> select (a strided sample of the gallery gives every query a score threshold that ~6 k + 100 rows reach; the similarity
 * pass appends the rows that reach it to a candidate buffer — no list kept; one wave per query then sorts out the k best; a query
 * whose candidate count fell outside [k, 2048] is redone by the streaming kernels: exact for any data); everything else, and
 * SLIC_TOPK_COLLECT=0, streams per-(query, gallery slice) k-slot heaps in LDS and merges them (SLIC_TOPK_COLLECT=1: collect for any k). */
size_t slic_cosine_topk_workspace_bytes(int Nq, int Ng, int k);
/* which of the two a call takes and the collect path's sample: out[0] = 1 collect / 0 streaming, out[1] = sample slices, out[2] = rows
 * per sample slice, out[3] = M (the threshold is the M-th best of the pooled sample), out[4] = gallery row step of the sample, out[5] = candidate
 * slots per query.  Introspection for tests and bench lines; no device work. */
int slic_cosine_topk_plan(int Nq, int Ng, int D, int k, int* out /* [6] */);
int slic_cosine_topk(const float* Qn, int Nq, const float* Gn, int Ng, int D, int k, int self_mask,
                     int32_t* out_idx, float* out_dist, void* workspace, void* stream);
/* merge W per-GPU result lists ([W, Nq, k] distances ascending + GLOBAL gallery indices, -1 = empty slot) into the
 * k nearest per query (distance ascending, ties -> lower index): the step after the all-gather when the gallery is
 * sharded by rows across GPUs */
int slic_topk_merge_lists(const float* pdist, const int32_t* pidx, int W, int Nq, int k, int32_t* out_idx,
                          float* out_dist, void* stream);
/* euclidean_distances(X, Y) as a dense [Nx, Ny] matrix (evaluate.py:216; validation-size inputs) */
int slic_pairwise_euclidean(const float* X, int Nx, const float* Y, int Ny, int D, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Memory-bank NCE (loss/NCE_loss.py:26-88 NCEAverage, :341-352 NCESoftmaxLoss): gather + dot + /T without
 * materialising the gathered rows, its backward wrt the features, the momentum bank update, and the
 * class-0 softmax cross-entropy.  idx / y are int64 (torch.long) as in the reference.
 * ---------------------------------------------------------------------------------------- */
/* out[b, j] = <bank[idx[b, j], :], f[b, :]> / T      idx: [B, K1], bank: [n_data, D], D % 4 == 0.
 * gathered (optional, [B*K1, D]): the rows as scored, written in the same pass — the training path needs them
 * for the backward because the bank is updated in place right after scoring (NCE_loss.py:73-86). */
int slic_nce_scores_fwd(const float* bank, const int64_t* idx, const float* f, int B, int K1, int D, float T,
                        float* out, float* gathered, void* stream);
/* df[b, :] = (1/T) sum_j dout[b, j] * bank[idx[b, j], :] */
int slic_nce_scores_bwd(const float* bank, const int64_t* idx, const float* dout, int B, int K1, int D, float T,
                        float* df, void* stream);
/* bank[y[b]] = normalise(momentum * bank[y[b]] + (1 - momentum) * f[b])  (NCE_loss.py:73-86) */
int slic_nce_bank_update(float* bank, const int64_t* y, const float* f, int B, int D, float momentum, void* stream);
/* loss = mean_b (logsumexp(x[b, :]) - x[b, 0]); lse/rowloss: [B] scratch kept for the backward */
int slic_softmax_ce0_fwd(const float* x, int B, int K1, float* lse, float* rowloss, float* loss, void* stream);
int slic_softmax_ce0_bwd(const float* x, const float* lse, int B, int K1, const float* gscale, float* dx, void* stream);
/* The contrastive step of online_train.py:175-190 — out_l, out_ab = contrast(feat_l, feat_ab, index); loss =
 * NCESoftmaxLoss(out_l) + NCESoftmaxLoss(out_ab) — as three launches (the separate entry points above: a dozen):
 *   slic_nce_fused_fwd   : scores[0] = <memory_l[idx], f_ab> / T (out_ab), scores[1] = <memory_ab[idx], f_l> / T (out_l), [2][B][K1];
 *                          rows [2][B][K1][D] = the bank rows as scored (the update below changes the banks before the backward);
 *                          part [2][B][SLIC_NCE_PARTS][2] = (max, sum-exp) of each of the SLIC_NCE_PARTS pieces a score row is split into
 *   slic_nce_fused_update: both banks' momentum update (NCE_loss.py:73-86; on duplicate labels the last row wins, every row computed
 *                          from the bank as it was); with part != NULL also lse / rowloss [2][B] = log-sum-exp of a score row and its
 *                          cross-entropy against class 0, and loss = mean_b rowloss[0][b] + mean_b rowloss[1][b]
 *   slic_nce_fused_bwd   : df[0] = d loss / d f_ab, df[1] = d loss / d f_l ([2][B][D]), scaled by *gscale (NULL = 1) */
#define SLIC_NCE_PARTS 16
int slic_nce_fused_fwd(const float* bank_l, const float* bank_ab, const float* f_l, const float* f_ab, const int64_t* idx, int B,
                       int K1, int D, float T, float* scores, float* rows, float* part, void* stream);
int slic_nce_fused_update(float* bank_l, float* bank_ab, const int64_t* y, const float* f_l, const float* f_ab, int B, int D,
                          float momentum, const float* part, const float* scores, int K1, float* lse, float* rowloss, float* loss,
                          void* stream);
int slic_nce_fused_bwd(const float* rows, const float* scores, const float* lse, int B, int K1, int D, float T,
                       const float* gscale, float* df, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SLIC_HIP_H */
