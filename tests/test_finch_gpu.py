"""GPU: FINCH with the first-neighbour search on the device vs goldens produced by importing the reference's
clustering/finch.py (tests/golden/make_goldens_finch.py)."""
import os

import numpy as np
import pytest
import torch



def _same_partition(a, b):
    """equal up to a relabelling"""
    pairs = set(zip(a.tolist(), b.tolist()))
    return len(pairs) == len(set(a.tolist())) == len(set(b.tolist()))


@pytest.mark.gpu
def test_finch_matches_reference_golden(gpu, golden_dir):
    from video_similarity_search_amd.clustering.finch import FINCH
    g = dict(np.load(os.path.join(golden_dir, "finch.npz")))
    c, num_clust, req_c = FINCH(g["X"], distance='cosine', verbose=False)
    assert list(num_clust) == list(g["num_clust"])
    assert c.shape == g["c"].shape
    for p in range(c.shape[1]):
        assert _same_partition(c[:, p], g["c"][:, p]), f"partition {p}"
    # connected_components numbers components in order of first appearance on both sides -> labels equal outright
    assert np.array_equal(c, g["c"])
    c2, nc2, req = FINCH(g["X"], req_clust=int(g["req_clust"]), distance='cosine', verbose=False)
    assert len(np.unique(req)) == int(g["req_clust"])
    assert _same_partition(req, g["req_c"])


@pytest.mark.gpu
def test_finch_deep_hierarchy_early_exit_and_req_clust(gpu, golden_dir):
    """three-level hierarchies from the imported reference: with the early-exit bound on link distances, without it,
    and a req_clust refinement that needs 10 single-pair merges below the last partition"""
    from video_similarity_search_amd.clustering.finch import FINCH
    g = dict(np.load(os.path.join(golden_dir, "finch.npz")))
    X = g["deep_X"]
    c, nc, _ = FINCH(X, distance='cosine', verbose=False)
    assert list(nc) == list(g["deep_num_clust"]) and np.array_equal(c, g["deep_c"])
    c, nc, _ = FINCH(torch.from_numpy(X).cuda(), distance='cosine', ensure_early_exit=False, verbose=False)     # resident input
    assert list(nc) == list(g["deep_noexit_num_clust"]) and np.array_equal(c, g["deep_noexit_c"])
    _, _, req = FINCH(X, req_clust=int(g["deep_req_clust"]), distance='cosine', ensure_early_exit=False, verbose=False)
    assert len(np.unique(req)) == int(g["deep_req_clust"])
    _, _, req = FINCH(X, req_clust=int(g["deep_req_clust"]), distance='cosine', verbose=False)
    assert _same_partition(req, g["deep_req_c"])


@pytest.mark.gpu
def test_finch_single_cluster_at_level_zero(gpu, golden_dir):
    """three mutually-near points: level 0 is already one cluster; the reference returns that single partition
    (the previous host driver died here with a negative column index)"""
    from video_similarity_search_amd.clustering.finch import FINCH
    g = dict(np.load(os.path.join(golden_dir, "finch.npz")))
    c, nc, req = FINCH(g["one_X"], distance='cosine', verbose=False)
    assert list(nc) == list(g["one_num_clust"]) == [1] and np.array_equal(c, g["one_c"]) and req is None
    c1, nc1, _ = FINCH(g["one_X"][:1], distance='cosine', verbose=False)             # one row
    assert list(nc1) == [1] and c1.shape == (1, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("t", [0, 1, 2])
def test_finch_weighted_early_exit_cut_matches_reference(gpu, golden_dir, t):
    """datasets on which the early-exit cut drops links AND the reference's adjacency weight (a mutual first-neighbour pair
    counts at 2 d, clustering/finch.py:38-40,44-45,144) decides the partition (the un-weighted cut gives
    other partitions here)"""
    from video_similarity_search_amd.clustering.finch import FINCH
    g = dict(np.load(os.path.join(golden_dir, "finch.npz")))
    X = g[f"cut{t}_X"]
    c, nc, _ = FINCH(X, distance='cosine', verbose=False)
    assert list(nc) == list(g[f"cut{t}_num_clust"]) and np.array_equal(c, g[f"cut{t}_c"])
    assert list(nc) != list(g[f"cut{t}_noexit_num_clust"])                         # the cut did drop links
    _, ncn, _ = FINCH(X, distance='cosine', ensure_early_exit=False, verbose=False)
    assert list(ncn) == list(g[f"cut{t}_noexit_num_clust"])
    _, _, req = FINCH(X, req_clust=10, distance='cosine', verbose=False)
    assert _same_partition(req, g[f"cut{t}_req_c"])


@pytest.mark.parametrize("name", ["cut0", "cut1", "cut2", "deep", "one"])
def test_finch_host_logic_vs_reference_golden_numpy_kernels(golden_dir, name):
    """the host side of clustering/finch.py (link list, weighted cut, hierarchy, stop rules) on a NumPy kernel provider
    passed as an argument: runs without a GPU, against the goldens of the imported reference"""
    from video_similarity_search_amd.clustering.finch import FINCH
    from finch_cpu_kernels import NumpyFinchKernels
    g = dict(np.load(os.path.join(golden_dir, "finch.npz")))
    c, nc, _ = FINCH(g[f"{name}_X"], distance='cosine', verbose=False, kernels=NumpyFinchKernels())
    assert list(nc) == list(g[f"{name}_num_clust"]) and np.array_equal(c, g[f"{name}_c"])
    if name.startswith("cut") or name == "deep":
        _, ncn, _ = FINCH(g[f"{name}_X"], distance='cosine', ensure_early_exit=False, verbose=False, kernels=NumpyFinchKernels())
        assert list(ncn) == list(g[f"{name}_noexit_num_clust"])
        rq = int(g["deep_req_clust"]) if name == "deep" else 10
        _, _, req = FINCH(g[f"{name}_X"], req_clust=rq, distance='cosine', verbose=False, kernels=NumpyFinchKernels())
        assert _same_partition(req, g[f"{name}_req_c"])


def test_finch_without_gpu_fails_loudly():
    from video_similarity_search_amd.clustering.finch import FINCH
    from video_similarity_search_amd._lib import SlicError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(SlicError):
        FINCH(np.zeros((4, 8), np.float32), verbose=False)


def test_link_pairs_definition():
    """the link set is exactly {i ~ j : j = k(i) or i = k(j) or k(i) = k(j)} (host-only helper)"""
    from video_similarity_search_amd.clustering.finch import link_pairs, components
    rng = np.random.default_rng(0)
    for n in (2, 7, 60):
        nn = np.array([rng.choice([j for j in range(n) if j != i]) for i in range(n)], np.int64)
        want = {(min(i, j), max(i, j)) for i in range(n) for j in range(n)
                if i != j and (nn[i] == j or nn[j] == i or nn[i] == nn[j])}
        a, b = link_pairs(nn)
        assert set(zip(a.tolist(), b.tolist())) == want and (a < b).all() and len(a) == len(want)
        lab, cnt = components(n, a, b)
        # numbered by smallest member
        firsts = [int(np.flatnonzero(lab == c)[0]) for c in range(cnt)]
        assert firsts == sorted(firsts)


@pytest.mark.gpu
def test_fit_cluster_finch_surface(gpu, golden_dir):
    from video_similarity_search_amd.clustering import fit_cluster
    g = dict(np.load(os.path.join(golden_dir, "finch.npz")))
    labels = fit_cluster(torch.from_numpy(g["X"]), method='finch', finch_partition=1)
    assert _same_partition(np.asarray(labels), g["c"][:, 1])
    from sklearn.metrics import normalized_mutual_info_score as nmi
    assert nmi(labels, g["z"]) > 0.99


@pytest.mark.gpu
def test_finch_beyond_flann_threshold_shape(gpu):
    """N > 70 000 (where the reference needs pyflann): exact first neighbours, sane partitions"""
    from video_similarity_search_amd.clustering.finch import FINCH, first_neighbours
    rng = np.random.default_rng(2)
    N, D = 80000, 64
    cent = rng.standard_normal((200, D)).astype(np.float32)
    z = rng.integers(0, 200, N)
    X = (cent[z] + 0.3 * rng.standard_normal((N, D))).astype(np.float32)
    Xd = torch.from_numpy(X).cuda()
    nn = first_neighbours(Xd)
    # spot-check the 1-NN of a few rows against brute force
    Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
    for i in (0, 17, 79999):
        s = Xn @ Xn[i]
        s[i] = -np.inf
        assert nn[i] == int(np.argmax(s))
    c, nc, _ = FINCH(Xd, distance='cosine', verbose=False)
    assert c.shape[0] == N and nc == sorted(nc, reverse=True) and nc[0] < N / 2
    from sklearn.metrics import normalized_mutual_info_score as nmi
    assert max(nmi(z, c[:, p]) for p in range(c.shape[1])) > 0.9
