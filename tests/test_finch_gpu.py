"""GPU: FINCH with the first-neighbour search on the device vs goldens produced by importing the reference's
clustering/finch.py (tests/golden/make_goldens_finch.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _same_partition(a, b):
    """equal up to a relabelling"""
    pairs = set(zip(a.tolist(), b.tolist()))
    return len(pairs) == len(set(a.tolist())) == len(set(b.tolist()))


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


def test_fit_cluster_finch_surface(gpu, golden_dir):
    from video_similarity_search_amd.clustering import fit_cluster
    g = dict(np.load(os.path.join(golden_dir, "finch.npz")))
    labels = fit_cluster(torch.from_numpy(g["X"]), method='finch', finch_partition=1)
    assert _same_partition(np.asarray(labels), g["c"][:, 1])
    from sklearn.metrics import normalized_mutual_info_score as nmi
    assert nmi(labels, g["z"]) > 0.99


def test_finch_beyond_flann_threshold_shape(gpu):
    """N > 70 000 (where the reference needs pyflann): exact 1-NN graph, sane partitions"""
    from video_similarity_search_amd.clustering.finch import clust_rank
    rng = np.random.default_rng(2)
    N, D = 80000, 64
    cent = rng.standard_normal((200, D)).astype(np.float32)
    X = (cent[rng.integers(0, 200, N)] + 0.3 * rng.standard_normal((N, D))).astype(np.float32)
    A, od = clust_rank(X)
    assert A.shape == (N, N) and od.nnz > 0
    # spot-check the 1-NN of a few rows against brute force
    Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
    from video_similarity_search_amd.evaluate import cosine_topk
    nn, _ = cosine_topk(X, None, k=1)
    nn = nn.view(-1).cpu().numpy()
    for i in (0, 17, 79999):
        s = Xn @ Xn[i]
        s[i] = -np.inf
        assert nn[i] == int(np.argmax(s))
