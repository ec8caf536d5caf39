"""
Generates tests/golden/kmeans_*.npz in the BUILD container (needs scikit-learn 1.7.2; the reference's
clustering/cluster_masks.py:70-71 calls sklearn.cluster.KMeans, which is where the arithmetic lives).
Inputs come from numpy.random.default_rng (PCG64, version-stable).  sklearn runs on ONE OpenMP thread so
its M-step sums are the ascending-row fp32 sums the oracle restates.

    python tests/golden/make_goldens_kmeans.py
"""
import os
import sys

import numpy as np
from sklearn.cluster import KMeans
from threadpoolctl import threadpool_limits

HERE = os.path.dirname(os.path.abspath(__file__))


def l2n(x):
    return (x / np.sqrt((x.astype(np.float64) ** 2).sum(1, keepdims=True))).astype(np.float32)


def sk_run(X, init, max_iter=300, tol=1e-4):
    with threadpool_limits(1):
        km = KMeans(n_clusters=init.shape[0], init=init.copy(), n_init=1, max_iter=max_iter, tol=tol,
                    algorithm="lloyd").fit(X)
    return km


def sk_trace(X, init, n_iter):
    """labels of the E-step that follows m centre updates, m = 1..n_iter-1 (a non-converged KMeans(max_iter=m)
    returns exactly those, _kmeans.py:736-748)"""
    import warnings
    tr = []
    for m in range(1, n_iter):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tr.append(sk_run(X, init, max_iter=m, tol=0.0).labels_.copy())
    return np.stack(tr) if tr else np.zeros((0, X.shape[0]), np.int32)


def case_unstructured():
    rng = np.random.default_rng(11)
    N, D, K = 4000, 32, 16
    X = l2n(rng.standard_normal((N, D)).astype(np.float32))
    init = X[rng.choice(N, K, replace=False)].copy()
    km = sk_run(X, init)
    return dict(X=X, init=init, labels=km.labels_.astype(np.int32), n_iter=km.n_iter_, inertia=km.inertia_,
                centers=km.cluster_centers_.astype(np.float32), trace=sk_trace(X, init, km.n_iter_))


def case_clustered_empty():
    """clustered data + an init with a far-away duplicate centre -> exactly one empty cluster in iteration 1,
    exercising _relocate_empty_clusters_dense with n_empty == 1 (the order-unambiguous case)"""
    rng = np.random.default_rng(12)
    N, D, K = 3000, 24, 10
    cent = l2n(rng.standard_normal((K, D)).astype(np.float32))
    z = rng.integers(0, K, N)
    X = l2n(cent[z] + 0.35 * rng.standard_normal((N, D)).astype(np.float32) / np.sqrt(D))
    init = X[rng.choice(N, K, replace=False)].copy()
    init[K - 1] = -8.0 * init[0]          # nobody is closest to this one
    km = sk_run(X, init)
    # confirm the branch fires: first E-step leaves cluster K-1 empty
    d = ((X[:, None, :] - init[None]) ** 2).sum(-1)
    assert (d.argmin(1) == K - 1).sum() == 0
    return dict(X=X, init=init, labels=km.labels_.astype(np.int32), n_iter=km.n_iter_, inertia=km.inertia_,
                centers=km.cluster_centers_.astype(np.float32), trace=sk_trace(X, init, km.n_iter_))


def case_d128():
    rng = np.random.default_rng(13)
    N, D, K = 2500, 128, 50
    X = l2n(rng.standard_normal((N, D)).astype(np.float32))
    init = X[rng.choice(N, K, replace=False)].copy()
    km = sk_run(X, init)
    return dict(X=X, init=init, labels=km.labels_.astype(np.int32), n_iter=km.n_iter_, inertia=km.inertia_,
                centers=km.cluster_centers_.astype(np.float32), trace=sk_trace(X, init, km.n_iter_))


def case_reference_call():
    """the reference-shaped call: np.random.seed(1) (cluster_masks.py:27) then KMeans(n_clusters=k, n_init=10)"""
    rng = np.random.default_rng(14)
    N, D, K = 1500, 16, 8
    cent = l2n(rng.standard_normal((K, D)).astype(np.float32))
    z = rng.integers(0, K, N)
    X = l2n(cent[z] + 0.5 * rng.standard_normal((N, D)).astype(np.float32) / np.sqrt(D))
    np.random.seed(1)
    with threadpool_limits(1):
        km = KMeans(n_clusters=K, n_init=10).fit(X)
    return dict(X=X, labels=km.labels_.astype(np.int32), inertia=km.inertia_, n_iter=km.n_iter_, true=z.astype(np.int32))


if __name__ == "__main__":
    import sklearn
    assert sklearn.__version__.startswith("1.7"), sklearn.__version__
    for name, fn in [("unstructured", case_unstructured), ("clustered_empty", case_clustered_empty),
                     ("d128", case_d128), ("reference_call", case_reference_call)]:
        d = fn()
        np.savez_compressed(os.path.join(HERE, f"kmeans_{name}.npz"), **d)
        print(name, {k: (v.shape if hasattr(v, "shape") and v.shape else v) for k, v in d.items() if k not in ("X",)})
