"""
ORACLE — TEST INFRASTRUCTURE ONLY.  NumPy restatement of the reference's retrieval arithmetic
(iic_retrieve_clips.py:275-314 topk_retrieval; evaluate.py:208-231, 287-307), whose distance computation is
sklearn.metrics.pairwise.cosine_distances (metrics/pairwise.py:1129: normalize both, 1 - X.Y^T, clip [0, 2]).
Pinned by tests/golden/retrieval.npz (generated with sklearn itself).
"""
import numpy as np


def normalize(X):
    X = np.asarray(X)
    n = np.sqrt((X.astype(np.float64) ** 2).sum(1, keepdims=True))
    n[n == 0] = 1.0
    return (X / n).astype(X.dtype)


def cosine_distances(X, Y=None):
    Xn = normalize(X)
    Yn = Xn if Y is None else normalize(Y)
    D = 1.0 - Xn @ Yn.T
    np.clip(D, 0, 2, out=D)
    if Y is None:
        np.fill_diagonal(D, 0.0)
    return D


def topk_retrieval(X_train, y_train, X_test, y_test, ks=(1, 5, 10, 20, 50)):
    d = cosine_distances(X_test, X_train)
    ind = np.argsort(d, kind="stable")
    out = {}
    for k in ks:
        lab = y_train[ind[:, :k]]
        out[k] = int((lab == y_test[:, None]).any(1).sum())
    return out, ind


def topk_acc_self(X, y, top_ks=(1, 5, 10, 20)):
    d = cosine_distances(X)
    np.fill_diagonal(d, np.inf)
    ind = np.argsort(d, kind="stable")[:, :top_ks[-1]]
    return np.array([(y[ind[:, :k]] == y[:, None]).any(1).mean() for k in top_ks]), ind
