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
