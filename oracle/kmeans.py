"""
ORACLE — TEST INFRASTRUCTURE ONLY (not the product path; only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this).

ctypes front-end of oracle/kmeans_oracle.c plus the reference-shaped wrappers:
  preprocess_features_kmeans  <- /root/reference/clustering/cluster_masks.py:30-34
  fit_cluster_kmeans          <- /root/reference/clustering/cluster_masks.py:64-71,92-98
                                 (-> sklearn KMeans.fit, _kmeans.py:1427-1540, n_init loop)
See the C file's header for the floating-point contract and parity status.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libslic_oracle.so")
        src = os.path.join(_HERE, "kmeans_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        L = ctypes.CDLL(so)
        L.slic_oracle_finalize.restype = ctypes.c_double
        L.slic_oracle_inertia.restype = ctypes.c_double
        L.slic_oracle_mean_var.restype = ctypes.c_double
        _LIB = L
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def num_threads():
    return int(_lib().slic_oracle_num_threads())


def row_sqnorm_chain(C):
    C = _f32(C)
    out = np.empty(C.shape[0], np.float32)
    _lib().slic_oracle_row_sqnorm_chain(_p(C), C.shape[0], C.shape[1], _p(out))
    return out


def assign(X, C, with_scores=False):
    X, C = _f32(X), _f32(C)
    N, D = X.shape
    labels = np.empty(N, np.int32)
    best = np.empty(N, np.float32)
    second = np.empty(N, np.float32)
    _lib().slic_oracle_assign(_p(X), ctypes.c_int64(N), D, _p(C), C.shape[0], _p(labels), _p(best), _p(second))
    return (labels, best, second) if with_scores else labels


def accumulate(X, labels, K, n_shards=1):
    X = _f32(X)
    labels = np.ascontiguousarray(labels, np.int32)
    N, D = X.shape
    sums = np.empty((K, D), np.float32)
    counts = np.empty(K, np.float32)
    _lib().slic_oracle_accumulate(_p(X), ctypes.c_int64(N), D, _p(labels), K, n_shards, _p(sums), _p(counts))
    return sums, counts


def dist_to_assigned(X, C, labels):
    X, C = _f32(X), _f32(C)
    labels = np.ascontiguousarray(labels, np.int32)
    out = np.empty(X.shape[0], np.float32)
    _lib().slic_oracle_dist_to_assigned(_p(X), ctypes.c_int64(X.shape[0]), X.shape[1], _p(C), _p(labels), _p(out))
    return out


def relocate_empty(X, C_old, labels, sums, counts):
    """in-place on sums/counts; returns n_empty"""
    X, C_old = _f32(X), _f32(C_old)
    labels = np.ascontiguousarray(labels, np.int32)
    assert sums.dtype == np.float32 and counts.dtype == np.float32
    return int(_lib().slic_oracle_relocate_empty(_p(X), ctypes.c_int64(X.shape[0]), X.shape[1], _p(C_old),
                                                 _p(labels), C_old.shape[0], _p(sums), _p(counts)))


def finalize(C_old, sums, counts):
    """returns (new_centres, shift[K], shift_tot)"""
    C_old = _f32(C_old)
    new = _f32(sums).copy()
    counts = _f32(counts)
    K, D = C_old.shape
    shift = np.empty(K, np.float32)
    tot = _lib().slic_oracle_finalize(_p(C_old), _p(new), _p(counts), K, D, _p(shift))
    return new, shift, float(tot)


def inertia(X, C, labels):
    X, C = _f32(X), _f32(C)
    labels = np.ascontiguousarray(labels, np.int32)
    return float(_lib().slic_oracle_inertia(_p(X), ctypes.c_int64(X.shape[0]), X.shape[1], _p(C), _p(labels)))


def col_mean(X):
    X = _f32(X)
    m = np.empty(X.shape[1], np.float32)
    _lib().slic_oracle_col_mean(_p(X), ctypes.c_int64(X.shape[0]), X.shape[1], _p(m))
    return m


def tolerance(Xc, tol):
    """sklearn _tolerance on the centred data (_kmeans.py:279-288)"""
    if tol == 0:
        return 0.0
    Xc = _f32(Xc)
    return float(_lib().slic_oracle_mean_var(_p(Xc), ctypes.c_int64(Xc.shape[0]), Xc.shape[1])) * tol


def lloyd(Xc, init, max_iter=300, tol_abs=0.0, n_shards=1, fixed_iters=False, trace=False):
    """_kmeans_single_lloyd on centred data.  Returns dict(labels, centers, inertia, n_iter, strict,
    n_relocations[, trace])."""
    Xc = _f32(Xc)
    C = _f32(init).copy()
    N, D = Xc.shape
    K = C.shape[0]
    labels = np.empty(N, np.int32)
    tr = np.full((max_iter, N), -1, np.int32) if trace else None
    strict = ctypes.c_int(0)
    nrel = ctypes.c_int(0)
    inert = ctypes.c_double(0)
    n_iter = _lib().slic_oracle_lloyd(_p(Xc), ctypes.c_int64(N), D, K, _p(C), max_iter, ctypes.c_double(tol_abs),
                                      n_shards, int(fixed_iters), _p(labels), _p(tr) if trace else None,
                                      ctypes.byref(strict), ctypes.byref(inert), ctypes.byref(nrel))
    out = dict(labels=labels, centers=C, inertia=inert.value, n_iter=int(n_iter), strict=bool(strict.value),
               n_relocations=nrel.value)
    if trace:
        out["trace"] = tr[:n_iter]
    return out


def preprocess_features_kmeans(data):
    """row L2 normalisation, no eps (cluster_masks.py:30-34: data / torch.norm(data, dim=1, keepdim=True))"""
    data = _f32(data)
    n = np.sqrt((data.astype(np.float64) ** 2).sum(1, keepdims=True)).astype(np.float32)
    return data / n


def kmeans_fit(X, init_list, max_iter=300, tol=1e-4, n_shards=1):
    """KMeans.fit with explicit initial centres (one per n_init run): centre, tol, runs, keep best inertia
    (_kmeans.py:1479-1537).  Returns the best run's dict with centres shifted back by the mean."""
    X = _f32(X)
    mean = col_mean(X)
    Xc = X - mean
    tol_abs = tolerance(Xc, tol)
    best = None
    for init in init_list:
        r = lloyd(Xc, _f32(init) - mean, max_iter=max_iter, tol_abs=tol_abs, n_shards=n_shards)
        if best is None or r["inertia"] < best["inertia"]:
            best = r
    best["centers"] = best["centers"] + mean
    best["tol_abs"] = tol_abs
    return best


def spherical_lloyd(X, init, max_iter=300, tol=1e-4):
    """PARITY UNPINNED.  spherecluster.SphericalKMeans (the package behind clustering/cluster_masks.py:73-77, pinned
    nowhere in the reference and absent from this image) restated from its published algorithm
    (_spherical_kmeans_single_lloyd): rows and initial centres L2-normalised, then repeat
    { euclidean assignment; centres = normalize(cluster means); stop when ||centres - centres_old||^2 <= tol (as given,
    not variance-scaled) }, a final relabelling if the last shift was non-zero.  float64 numpy; empty clusters keep a zero
    centre until relocation, which the small test cases never hit.  Returns (labels, centres, n_iter)."""
    X = np.asarray(X, np.float64)
    X = X / np.linalg.norm(X, axis=1, keepdims=True)
    C = np.asarray(init, np.float64)
    K = C.shape[0]
    n_iter, shift = 0, 1.0
    labels = None
    for it in range(max_iter):
        d = (X * X).sum(1)[:, None] - 2.0 * X @ C.T + (C * C).sum(1)[None, :]
        labels = d.argmin(1)
        Cn = np.zeros_like(C)
        np.add.at(Cn, labels, X)
        cnt = np.bincount(labels, minlength=K).astype(np.float64)
        Cn = Cn / np.maximum(cnt, 1)[:, None]
        nrm = np.linalg.norm(Cn, axis=1, keepdims=True)
        Cn = np.where(nrm > 0, Cn / np.where(nrm > 0, nrm, 1), Cn)
        shift = float(((Cn - C) ** 2).sum())
        C = Cn
        n_iter = it + 1
        if shift <= tol:
            break
    if shift > 0:
        d = (X * X).sum(1)[:, None] - 2.0 * X @ C.T + (C * C).sum(1)[None, :]
        labels = d.argmin(1)
    return labels.astype(np.int32), C, n_iter
