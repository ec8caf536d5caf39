"""GPU parity: HIP k-means (through the C ABI) vs the CPU oracle and the sklearn goldens.
Bar: labels bit-exact (every iteration), centres / scores bit-exact, n_iter and convergence flag equal."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, f"kmeans_{name}.npz")))


def _assign_gpu(X, C):
    from video_similarity_search_amd import _lib
    from video_similarity_search_amd._lib import call, ptr, stream
    lib = _lib.load()
    Xd, Cd = torch.from_numpy(X).cuda(), torch.from_numpy(C).cuda()
    N, D = X.shape
    K = C.shape[0]
    cn = torch.empty(K, device="cuda")
    lab = torch.empty(N, dtype=torch.int32, device="cuda")
    best = torch.empty(N, device="cuda")
    ws = torch.empty(lib.slic_kmeans_assign_workspace_bytes(N, K), dtype=torch.uint8, device="cuda")
    call("slic_kmeans_cnorm", ptr(Cd), K, D, D, ptr(cn), stream())
    call("slic_kmeans_assign", ptr(Xd), N, D, D, ptr(Cd), K, D, ptr(cn), ptr(lab), None, None, ptr(best), ptr(ws), stream())
    torch.cuda.synchronize()
    return lab.cpu().numpy(), best.cpu().numpy(), cn.cpu().numpy()


@pytest.mark.parametrize("N,D,K", [(1000, 64, 37), (257, 8, 3), (4096, 512, 500), (130, 40, 129), (5, 16, 1)])
def test_assign_scores_bit_exact(gpu, N, D, K):
    """the MFMA k-chain == the oracle's fmaf chain: winning scores and labels agree to the bit"""
    from oracle import kmeans as ok
    rng = np.random.default_rng(N + D + K)
    X = rng.standard_normal((N, D)).astype(np.float32)
    C = rng.standard_normal((K, D)).astype(np.float32)
    lab, best, cn = _assign_gpu(X, C)
    olab, obest, _ = ok.assign(X, C, with_scores=True)
    assert np.array_equal(cn.view(np.uint32), ok.row_sqnorm_chain(C).view(np.uint32))
    assert np.array_equal(best.view(np.uint32), obest.view(np.uint32)), np.abs(best - obest).max()
    assert np.array_equal(lab, olab)


def _assign_perm_gpu(X, C):
    """the product E-step: k8-permuted operands -> slic_kmeans_assign_perm (km_assign_creg / km_assign_dma) -> km_combine"""
    from video_similarity_search_amd import _lib
    from video_similarity_search_amd._lib import call, ptr, stream
    N, D = X.shape
    K = C.shape[0]
    lib = _lib.load()
    Xd, Cd = torch.from_numpy(X).cuda(), torch.from_numpy(C).cuda()
    Xp, Cp = torch.empty_like(Xd), torch.empty_like(Cd)
    call("slic_kmeans_permute_k8", ptr(Xd), N, D, D, ptr(Xp), D, stream())
    call("slic_kmeans_permute_k8", ptr(Cd), K, D, D, ptr(Cp), D, stream())
    cn = torch.empty(K, device="cuda")
    call("slic_kmeans_cnorm", ptr(Cd), K, D, D, ptr(cn), stream())
    lab = torch.empty(N, dtype=torch.int32, device="cuda")
    best = torch.empty(N, dtype=torch.float32, device="cuda")
    ws = torch.empty(lib.slic_kmeans_assign_workspace_bytes(N, K), dtype=torch.uint8, device="cuda")
    call("slic_kmeans_assign_perm", ptr(Xp), N, D, D, ptr(Cp), K, D, ptr(cn), ptr(lab), None, None, ptr(best), ptr(ws), stream())
    torch.cuda.synchronize()
    return lab.cpu().numpy(), best.cpu().numpy()


# centroids in registers with 16 / 8 / 4 k-tiles (D <= 512 / 256 / 128; ragged last tile, K not a multiple of 128 or 32),
# and the shapes that fall back to the 128 x 64 tile kernel (few points, K far below a multiple of 128, D > 512)
@pytest.mark.parametrize("N,D,K", [(40000, 512, 500), (33003, 200, 384), (70001, 128, 250), (2000, 512, 500),
                                   (40000, 64, 130), (9000, 520, 500)])
def test_assign_perm_scores_bit_exact(gpu, N, D, K):
    from oracle import kmeans as ok
    rng = np.random.default_rng(N + D + K)
    X = rng.standard_normal((N, D)).astype(np.float32)
    C = rng.standard_normal((K, D)).astype(np.float32)
    C[K // 2] = C[3]                                                 # an exact tie: the lower index must win
    lab, best = _assign_perm_gpu(X, C)
    olab, obest, _ = ok.assign(X, C, with_scores=True)
    assert np.array_equal(best.view(np.uint32), obest.view(np.uint32)), np.abs(best - obest).max()
    assert np.array_equal(lab, olab)
    assert not np.any(lab == K // 2)


def test_assign_ties_first_index(gpu):
    X = np.ones((300, 16), np.float32)
    C = np.zeros((140, 16), np.float32)
    C[7] = 1.0
    C[135] = 1.0                      # exact tie between 7 and 135 -> 7 (first index)
    lab, _, _ = _assign_gpu(X, C)
    assert (lab == 7).all()
    lab0, _, _ = _assign_gpu(np.zeros((10, 16), np.float32), np.zeros((200, 16), np.float32))
    assert (lab0 == 0).all()


@pytest.mark.parametrize("name", ["unstructured", "clustered_empty", "d128"])
def test_fit_matches_oracle_and_golden(gpu, golden_dir, name):
    from oracle import kmeans as ok
    from video_similarity_search_amd.clustering import KMeans
    g = _load(golden_dir, name)
    X, init = g["X"], g["init"]
    km = KMeans(n_clusters=init.shape[0], init=init, n_init=1, trace=True).fit(torch.from_numpy(X))
    mean = ok.col_mean(X)
    Xc = X - mean
    tol_abs = ok.tolerance(Xc, 1e-4)
    r = ok.lloyd(Xc, init - mean, tol_abs=tol_abs, trace=True)
    assert km.tol_abs_ == pytest.approx(tol_abs, rel=1e-12)
    assert km.n_iter_ == r["n_iter"] == int(g["n_iter"])
    assert km.strict_ == r["strict"]
    assert km.trace_.shape == r["trace"].shape
    for it in range(r["n_iter"]):
        bad = np.nonzero(km.trace_[it] != r["trace"][it])[0]
        assert bad.size == 0, f"iteration {it}: {bad.size} label mismatches"
    assert np.array_equal(km.labels_, r["labels"])
    assert np.array_equal(km.labels_, g["labels"])           # and the sklearn golden itself
    assert km.labels_.dtype == np.int32
    np.testing.assert_allclose(km.cluster_centers_, r["centers"] + mean, rtol=0, atol=0)
    assert km.inertia_ == pytest.approx(r["inertia"], rel=1e-12)
    if name == "clustered_empty":
        assert km.n_relocations_ >= 1 and km.n_relocations_ == r["n_relocations"]


def test_accumulate_bit_exact_and_ragged(gpu):
    from oracle import kmeans as ok
    from video_similarity_search_amd.clustering.kmeans_hip import HipKernels
    rng = np.random.default_rng(5)
    for N, D, K in [(1000, 16, 7), (70001, 128, 300), (3, 8, 5)]:
        X = rng.standard_normal((N, D)).astype(np.float32)
        lab = rng.integers(0, K, N).astype(np.int32)
        lab[lab == 2] = 3                                    # an empty cluster
        Xd, ld = torch.from_numpy(X).cuda(), torch.from_numpy(lab).cuda()
        sums = torch.empty(K * D, device="cuda")
        counts = torch.empty(K, device="cuda")
        HipKernels().accumulate(Xd, ld, K, sums, counts)
        s, c = ok.accumulate(X, lab, K)
        assert np.array_equal(counts.cpu().numpy(), c)
        assert np.array_equal(sums.cpu().numpy().reshape(K, D).view(np.uint32), s.view(np.uint32))


def test_full_size_three_iterations(gpu):
    """BASELINE config 3 shape on one GPU: 100k x 512, K = 500, explicit init, labels of every iteration"""
    from oracle import kmeans as ok
    from video_similarity_search_amd.clustering import KMeans
    rng = np.random.default_rng(1)
    N, D, K = 100000, 512, 500
    X = rng.standard_normal((N, D)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    init = X[rng.choice(N, K, replace=False)].copy()
    km = KMeans(n_clusters=K, init=init, n_init=1, max_iter=3, tol=0.0, fixed_iters=True, trace=True).fit(torch.from_numpy(X))
    mean = ok.col_mean(X)
    r = ok.lloyd(X - mean, init - mean, max_iter=3, tol_abs=0.0, fixed_iters=True, trace=True)
    for it in range(3):
        assert np.array_equal(km.trace_[it], r["trace"][it]), f"iteration {it}"
    assert np.array_equal(km.labels_, r["labels"])
    # size-independent properties: every label in range, counts sum to N, inertia decreased by the updates
    assert km.labels_.min() >= 0 and km.labels_.max() < K
    assert km.inertia_ == pytest.approx(r["inertia"], rel=1e-10)


def test_l2norm_rows(gpu):
    from video_similarity_search_amd.clustering import preprocess_features_kmeans
    from oracle import kmeans as ok
    rng = np.random.default_rng(2)
    X = (rng.standard_normal((1234, 128)) * 3).astype(np.float32)
    out = preprocess_features_kmeans(torch.from_numpy(X)).cpu().numpy()
    np.testing.assert_allclose(out, ok.preprocess_features_kmeans(X), rtol=2e-7, atol=0)
    ref = (torch.from_numpy(X) / torch.norm(torch.from_numpy(X), dim=1, keepdim=True)).numpy()   # the reference's line
    np.testing.assert_allclose(out, ref, rtol=3e-7, atol=0)


def test_fit_cluster_reference_call(gpu, golden_dir):
    """fit_cluster(embeddings, 'kmeans', k) as online_train.py:625 calls it — np.random.seed(1) (cluster_masks.py:27), then
    KMeans(n_clusters=k, n_init=10) with k-means++ — against sklearn's golden for that very call: the host side draws from NumPy's global
    RNG exactly as sklearn 1.7.2 does (first centre by choice, n_local_trials uniforms per further centre, ten initialisations in
    sequence), so the SAME rows are picked and the labels — cluster numbering included — are sklearn's, not merely a partition of
    equal quality (tests/test_oracle_kmeans.py pins the host logic on the CPU kernels; here the device kernels compute the distances)"""
    from video_similarity_search_amd.clustering import fit_cluster
    g = _load(golden_dir, "reference_call")
    np.random.seed(1)
    labels = fit_cluster(torch.from_numpy(g["X"]), method="kmeans", k=8, l2normalize=True)
    assert labels.shape == (1500,) and labels.dtype == np.int32
    km = fit_cluster.last_model
    assert len(set(km.init_indices_.tolist())) == 8
    assert np.array_equal(labels, g["labels"]), int((labels != g["labels"]).sum())
    assert km.n_iter_ == int(g["n_iter"]) and km.inertia_ == pytest.approx(float(g["inertia"]), rel=1e-6)


def test_kmeanspp_helpers(gpu):
    from video_similarity_search_amd.clustering.kmeans_hip import HipKernels
    rng = np.random.default_rng(9)
    N, D, T = 5000, 64, 6
    X = rng.standard_normal((N, D)).astype(np.float32)
    Xd = torch.from_numpy(X).cuda()
    k = HipKernels()
    cand = torch.tensor([3, 77, 1000, 4999, 0, 2500], dtype=torch.int32, device="cuda")
    closest = torch.from_numpy(rng.random(N).astype(np.float32) * 100).cuda()
    nd = torch.empty(T, N, device="cuda")
    pot = torch.empty(T, dtype=torch.float64, device="cuda")
    k.kpp_step(Xd, cand, T, closest, nd, pot)
    ref = np.minimum(((X[None, :, :].astype(np.float64) - X[cand.cpu().numpy()][:, None, :]) ** 2).sum(-1), closest.cpu().numpy()[None])
    np.testing.assert_allclose(nd.cpu().numpy(), ref, rtol=2e-6, atol=1e-5)
    np.testing.assert_allclose(pot.cpu().numpy(), ref.sum(1), rtol=1e-6)
    v = torch.from_numpy(rng.random(N).astype(np.float32)).cuda()
    cs = np.cumsum(v.cpu().numpy().astype(np.float64))
    vals_h = np.array([0.0, cs[10], cs[-1] * 0.5, cs[-1], cs[-1] * 2, 1e-9])
    vals = torch.from_numpy(vals_h).cuda()
    idx = torch.empty(len(vals_h), dtype=torch.int32, device="cuda")
    k.cumsum_search(v, vals, len(vals_h), idx)
    exp = np.clip(np.searchsorted(cs, vals_h), None, N - 1)
    got = idx.cpu().numpy()
    assert np.all(np.abs(got - exp) <= 1), (got, exp)      # chunked double sums vs np.cumsum: boundary +-1


@pytest.mark.parametrize("N,D,K,R", [(6000, 64, 24, 4), (20000, 128, 100, 10), (30000, 512, 500, 10), (3000, 32, 6, 3)])
def test_kmeanspp_batch_equals_sequential(gpu, monkeypatch, N, D, K, R):
    """KMeans(n_init = R): the R k-means++ seedings taken in lock-step (slic_kmeanspp_run_batch: one pass over X per centre for all
    runs, 1-3 MFMA row tiles of candidates) pick exactly the rows the run-by-run loop picks from the same RNG stream, and the fit
    ends on the same labels, centres and inertia"""
    from video_similarity_search_amd.clustering import KMeans
    rng = np.random.default_rng(N + K)
    cen = rng.standard_normal((K, D)).astype(np.float32)
    X = (cen[rng.integers(0, K, N)] + 0.7 * rng.standard_normal((N, D))).astype(np.float32)
    Xd = torch.from_numpy(X).cuda()
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SLIC_KPP_BATCH", mode)
        km = KMeans(n_clusters=K, n_init=R, max_iter=5, random_state=3).fit(Xd)
        res[mode] = (np.stack(km.init_indices_log_), km.labels_.copy(), km.cluster_centers_.copy(), km.inertia_)
    assert res["1"][0].shape == (R, K)
    assert np.array_equal(res["1"][0], res["0"][0])
    assert len(set(map(tuple, res["1"][0]))) == R                    # the runs differ from each other (one stream, not R copies)
    assert np.array_equal(res["1"][1], res["0"][1]) and np.array_equal(res["1"][2], res["0"][2]) and res["1"][3] == res["0"][3]


@pytest.mark.parametrize("N,D,K", [(6000, 64, 24), (1000, 40, 8), (333, 8, 5)])
def test_kmeanspp_run_matches_stepwise(gpu, N, D, K):
    """the single-sequence k-means++ (slic_kmeanspp_run) picks the rows the stepwise host loop picks from the same
    uniform draws (distances are summed in a different order, so the data is chosen without near-ties), and both follow
    _kmeans_plusplus restated in float64 numpy"""
    from video_similarity_search_amd.clustering.kmeans_hip import HipKernels
    rng = np.random.default_rng(21)
    T = 2 + int(np.log(K))
    cen = rng.standard_normal((K, D)) * 3
    X = (cen[rng.integers(0, K, N)] + 0.3 * rng.standard_normal((N, D))).astype(np.float32)
    u = rng.random((K - 1, T))
    first = 1234 % N
    # float64 restatement (sklearn/cluster/_kmeans.py:174-277)
    Xd64 = X.astype(np.float64)
    closest = ((Xd64 - Xd64[first]) ** 2).sum(1)
    ref = [first]
    for c in range(1, K):
        cand = np.clip(np.searchsorted(np.cumsum(closest), u[c - 1] * closest.sum()), None, N - 1)
        d = np.minimum(((Xd64[None] - Xd64[cand][:, None]) ** 2).sum(-1), closest[None])
        b = int(np.argmin(d.sum(1)))
        closest = d[b]
        ref.append(int(cand[b]))
    k = HipKernels()
    Xg = torch.from_numpy(X).cuda()
    idx = torch.empty(K, dtype=torch.int32, device="cuda")
    k.kpp_run(Xg, first, K, T, torch.from_numpy(u).cuda(), idx)
    got = idx.cpu().numpy().tolist()
    assert got[0] == first and len(set(got)) == K
    assert got == ref
    # matrix-pipe distances (|x|^2 + |c|^2 - 2 x.c on the k-permuted copy): same picks on this well-separated data
    Xp, xn = torch.empty_like(Xg), torch.empty(N, device="cuda")
    k.permute_k8(Xg, Xp)
    k.cnorm(Xg, xn)
    idx2 = torch.empty(K, dtype=torch.int32, device="cuda")
    k.kpp_run(Xg, first, K, T, torch.from_numpy(u).cuda(), idx2, Xp, xn)
    assert idx2.cpu().numpy().tolist() == ref


def test_spherical_kmeans_matches_restatement(gpu):
    """spherical k-means (clustering/cluster_masks.py:73-77).  Its oracle (spherecluster) is not
    vendored, so parity is UNPINNED: the HIP path is checked against oracle.kmeans.spherical_lloyd (the published
    algorithm in float64) from the same explicit unit-norm init, and against the definition's invariants."""
    from oracle import kmeans as ok
    from video_similarity_search_amd.clustering import KMeans, fit_cluster
    rng = np.random.default_rng(31)
    N, D, K = 4000, 64, 12
    cen = rng.standard_normal((K, D))
    cen /= np.linalg.norm(cen, axis=1, keepdims=True)
    z = rng.integers(0, K, N)
    X = (cen[z] + 0.25 * rng.standard_normal((N, D)) / np.sqrt(D)).astype(np.float32) * rng.uniform(0.5, 3.0, (N, 1)).astype(np.float32)
    Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
    # one seed per true cluster, from two mixed points each, so no cluster runs empty (relocation is not part of the restatement)
    init = np.stack([Xn[np.flatnonzero(z == j)[:2]].sum(0) for j in range(K)])
    init = (init / np.linalg.norm(init, axis=1, keepdims=True)).astype(np.float32)
    perm = rng.permutation(K)
    init = init[perm]
    ref_labels, ref_C, ref_iter = ok.spherical_lloyd(X, init)
    km = KMeans(n_clusters=K, init=init, n_init=1, spherical=True).fit(torch.from_numpy(X).cuda())
    assert np.array_equal(km.labels_, ref_labels)
    assert km.n_iter_ == ref_iter
    np.testing.assert_allclose(km.cluster_centers_, ref_C, atol=2e-6, rtol=0)
    np.testing.assert_allclose(np.linalg.norm(km.cluster_centers_, axis=1), 1.0, atol=1e-6)
    assert np.array_equal(km.labels_, (Xn @ km.cluster_centers_.T).argmax(1))          # nearest centre = largest cosine
    labels = fit_cluster(torch.from_numpy(X), method="spherical_kmeans", k=K)
    from sklearn.metrics import normalized_mutual_info_score as nmi
    assert labels.shape == (N,) and nmi(labels, ref_labels) > 0.9


@pytest.mark.parametrize("f64", [False, True])
def test_sharded_iteration_halves_vs_cpu_twin(gpu, f64):
    """slic_kmeans_lloyd_local / slic_kmeans_lloyd_global (the sharded iteration around its ONE collective) on one GPU with W = 3
    synthetic shards: the payload of each shard == the oracle's E-step + ordered M-step on that shard's rows, and the combine +
    averaging (+ an empty cluster: the copy-the-biggest path) == the CPU twin (tests/kmeans_cpu_kernels.py) bit for bit"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from kmeans_cpu_kernels import OracleKernels
    from oracle import kmeans as ok
    from video_similarity_search_amd.clustering.kmeans_hip import HipKernels
    hk, okk = HipKernels(), OracleKernels()
    rng = np.random.default_rng(11)
    N, D, K, W = 3 * 700, 40, 150, 3
    X = rng.standard_normal((N, D)).astype(np.float32)
    C = X[rng.choice(N, K, replace=False)].copy()
    C[7] = 50.0                                              # a centre no row is closest to: empty cluster
    dt = torch.float64 if f64 else torch.float32
    PL = K * D + K + 2
    Cd = torch.from_numpy(C).cuda()
    Cp = torch.empty_like(Cd)
    cn = torch.empty(K, device="cuda")
    hk.permute_k8(Cd, Cp)
    hk.cnorm(Cd, cn)
    parts_g, parts_c, labs = [], [], []
    for s in range(W):
        Xs = torch.from_numpy(X[s * 700:(s + 1) * 700]).cuda()
        Xp = torch.empty_like(Xs)
        hk.permute_k8(Xs, Xp)
        lab = torch.empty(700, dtype=torch.int32, device="cuda")
        old = torch.full((700,), -1, dtype=torch.int32, device="cuda")
        pay = torch.empty(1, PL, dtype=dt, device="cuda")
        hk.lloyd_local(Xs, Xp, Cd, Cp, cn, lab, old, pay)
        payc = torch.empty(1, PL, dtype=dt)
        labc = torch.empty(700, dtype=torch.int32)
        okk.lloyd_local(Xs.cpu(), None, torch.from_numpy(C), None, None, labc, old.cpu(), payc)
        assert torch.equal(lab.cpu(), labc)
        assert torch.equal(pay.cpu(), payc), s
        assert pay[0, K * D + K].item() == 700 and pay[0, K * D + K + 1].item() == 0      # every label changed from -1
        parts_g.append(pay)
        parts_c.append(payc)
        labs.append(labc)
    if f64:
        pg, pc = sum(parts_g[1:], parts_g[0].clone()), sum(parts_c[1:], parts_c[0].clone())      # what the all-reduce hands back
    else:
        pg, pc = torch.cat(parts_g), torch.cat(parts_c)
    out_g = [torch.empty(K * D, device="cuda"), torch.empty(K, device="cuda"), torch.empty(K, D, device="cuda"),
             torch.empty(K, D, device="cuda"), torch.empty(K, device="cuda"), torch.empty(K, device="cuda"),
             torch.empty(4, dtype=torch.float64, device="cuda")]
    hk.lloyd_global(pg, Cd, *out_g)
    out_c = [torch.empty(K * D), torch.empty(K), torch.empty(K, D), None, torch.empty(K), torch.empty(K),
             torch.empty(4, dtype=torch.float64)]
    okk.lloyd_global(pc, torch.from_numpy(C), *out_c)
    torch.cuda.synchronize()
    sums_ref, counts_ref = ok.accumulate(X, np.concatenate([l.numpy() for l in labs]), K, -W if f64 else W)
    assert np.array_equal(out_g[0].cpu().numpy().reshape(K, D), sums_ref) and np.array_equal(out_g[1].cpu().numpy(), counts_ref)
    assert counts_ref[7] == 0
    for i in (0, 1, 2, 4, 5):
        assert torch.equal(out_g[i].cpu(), out_c[i]), i
    Cp_new = torch.empty(K, D, device="cuda")
    hk.permute_k8(out_g[2], Cp_new)
    assert torch.equal(Cp_new, out_g[3])
    sg, sc = out_g[6].cpu().numpy(), out_c[6].numpy()
    assert sg[0] == pytest.approx(sc[0], rel=1e-12) and sg[1] == sc[1] == 1.0 and sg[2] == sc[2] == N


def test_slic_comm_one_rank_allreduce(gpu):
    """slic_comm_unique_id / slic_comm_create / slic_allreduce_f32 / _f64 / slic_comm_destroy on a communicator of one rank (the
    widest a one-GPU box allows): the sum over one rank is the buffer itself, on the caller's stream"""
    import ctypes
    from video_similarity_search_amd import _lib
    from video_similarity_search_amd._lib import call, ptr, stream
    lib = _lib.load()
    buf = (ctypes.c_ubyte * 128)()
    _lib.check(lib.slic_comm_unique_id(buf), "slic_comm_unique_id")
    assert any(buf)
    comm = ctypes.c_void_p()
    _lib.check(lib.slic_comm_create(bytes(buf), 1, 0, ctypes.byref(comm)), "slic_comm_create")
    assert comm.value
    a = torch.arange(1000, dtype=torch.float32, device="cuda") * 0.5
    b = torch.arange(777, dtype=torch.float64, device="cuda") / 3
    a0, b0 = a.clone(), b.clone()
    call("slic_allreduce_f32", comm, ptr(a), a.numel(), stream())
    call("slic_allreduce_f64", comm, ptr(b), b.numel(), stream())
    torch.cuda.synchronize()
    assert torch.equal(a, a0) and torch.equal(b, b0)
    assert lib.slic_allreduce_f32(comm, None, 4, None) != 0 and b"bad args" in lib.slic_last_error()
    _lib.check(lib.slic_comm_wait(comm, stream(), 5000), "slic_comm_wait")          # everything enqueued has run: returns at once
    _lib.check(lib.slic_comm_destroy(comm), "slic_comm_destroy")


def test_slic_comm_missing_peer_times_out(gpu):
    """a communicator whose peer never shows up (world = 2 joined by ONE process — what a rank lost before the rendezvous looks
    like) must not hang: slic_comm_create_timeout gives up after its deadline with SLIC_ETIMEOUT, the communicator is aborted and
    the process goes on (here: a fresh one-rank communicator works right after)"""
    import ctypes
    import time
    from video_similarity_search_amd import _lib
    from video_similarity_search_amd._lib import call, ptr, stream
    lib = _lib.load()
    buf = (ctypes.c_ubyte * 128)()
    _lib.check(lib.slic_comm_unique_id(buf), "slic_comm_unique_id")
    comm = ctypes.c_void_p()
    t0 = time.time()
    rc = lib.slic_comm_create_timeout(bytes(buf), 2, 0, 3000, ctypes.byref(comm))
    dt = time.time() - t0
    assert rc == -4 and not comm.value, (rc, lib.slic_last_error())               # SLIC_ETIMEOUT
    assert 2.5 <= dt < 30 and b"aborted" in lib.slic_last_error(), (dt, lib.slic_last_error())
    # the process is intact: a one-rank communicator with a deadline, an all-reduce, a bounded wait, an explicit abort
    _lib.check(lib.slic_comm_unique_id(buf), "slic_comm_unique_id")
    _lib.check(lib.slic_comm_create_timeout(bytes(buf), 1, 0, 10000, ctypes.byref(comm)), "slic_comm_create_timeout")
    b = torch.arange(4096, dtype=torch.float64, device="cuda") / 7
    b0 = b.clone()
    call("slic_allreduce_f64", comm, ptr(b), b.numel(), stream())
    _lib.check(lib.slic_comm_wait(comm, stream(), 10000), "slic_comm_wait")
    assert torch.equal(b, b0)
    _lib.check(lib.slic_comm_abort(comm), "slic_comm_abort")


def test_sharded_kmeans_through_the_library_communicator(gpu, monkeypatch):
    """SLIC_KMEANS_COMM=slic (opt-in): the sharded iteration's one all-reduce through slic_allreduce_f64 on a one-rank RCCL group,
    every status read behind slic_comm_wait's deadline — same labels as the default route through torch.distributed"""
    import subprocess
    import sys
    from conftest import ROOT
    code = (
        "import os, numpy as np, torch, torch.distributed as dist\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29631', RANK='0', WORLD_SIZE='1')\n"
        "dist.init_process_group('nccl'); torch.cuda.set_device(0)\n"
        "from video_similarity_search_amd.clustering import KMeans\n"
        "rng = np.random.default_rng(3); X = rng.standard_normal((4096, 64)).astype(np.float32)\n"
        "init = X[rng.choice(4096, 16, replace=False)].copy(); Xd = torch.from_numpy(X).cuda()\n"
        "out = []\n"
        "for route in ('torch', 'slic'):\n"
        "    os.environ['SLIC_KMEANS_COMM'] = route\n"
        "    km = KMeans(16, init=init, n_init=1, max_iter=15, process_group=dist.group.WORLD).fit(Xd)\n"
        "    out.append((km.labels_.copy(), km.n_iter_, km.cluster_centers_.copy()))\n"
        "assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1] and np.array_equal(out[0][2], out[1][2])\n"
        "dist.destroy_process_group(); print('routes agree', out[0][1])\n")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "routes agree" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_scan_accumulate_equals_counting_sort(gpu, monkeypatch):
    """the M-step sums of a small shard by label scan (km_scan_accumulate: one launch, every cluster's workgroup finds its rows itself) against
    the counting sort + ordered sums (SLIC_KM_SCAN=0): the same row order, so the SAME sums bit for bit and the same counts — random shapes up
    to the 32 768-row limit: row counts off every 16 / 1024 boundary, empty clusters, one giant cluster, D from 4 to 512, against a float64
    reference too; and the sharded local half (fp64 payload, permuted source) through both"""
    from video_similarity_search_amd.clustering.kmeans_hip import HipKernels
    k = HipKernels()
    rng = np.random.default_rng(31)
    cases = [(1, 1, 4), (17, 3, 8), (1000, 500, 512), (12500, 500, 512), (32768, 37, 128), (32767, 1000, 64), (4097, 130, 200), (20000, 2, 512),
             (12345, 500, 40)]
    for N, K, D in cases:
        X = torch.from_numpy(rng.standard_normal((N, D)).astype(np.float32)).cuda()
        lab = rng.integers(0, K, N).astype(np.int32)
        if K > 3:
            lab[lab == 1] = 0                                       # an empty cluster and a crowded one
        if N == 20000:
            lab[:] = 1                                              # every row in one cluster
        labd = torch.from_numpy(lab).cuda()
        out = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("SLIC_KM_SCAN", mode)
            sums = torch.full((K, D), -7.0, device="cuda")
            counts = torch.full((K,), -7.0, device="cuda")
            k.accumulate(X, labd, K, sums, counts)
            out[mode] = (sums.cpu().numpy(), counts.cpu().numpy())
        monkeypatch.delenv("SLIC_KM_SCAN")
        assert np.array_equal(out["1"][0].view(np.uint32), out["0"][0].view(np.uint32)), (N, K, D)
        assert np.array_equal(out["1"][1], out["0"][1]) and np.array_equal(out["1"][1], np.bincount(lab, minlength=K).astype(np.float32))
        ref = np.zeros((K, D))
        np.add.at(ref, lab, X.cpu().numpy().astype(np.float64))
        np.testing.assert_allclose(out["1"][0], ref, rtol=2e-5, atol=2e-4 * np.sqrt(max(1, N / K)))
