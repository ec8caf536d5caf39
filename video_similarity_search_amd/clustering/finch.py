"""
FINCH (Sarfraz et al., CVPR 2019) with the first-neighbour search on the GPU — the clustering method every
shipped SLIC config selects (ITERCLUSTER.METHOD: finch).

Mirrors the reference's clustering/finch.py (same function names and return values):
    clust_rank(mat, initial_rank, distance)      <- finch.py:22-47
    get_clust(a, orig_dist, min_sim)             <- finch.py:50-55
    cool_mean(M, u) / get_merge(c, u, data)      <- finch.py:58-82
    FINCH(data, initial_rank, req_clust, distance, ensure_early_exit, verbose) -> (c, num_clust, req_c)   <- finch.py:108-178

What changes:
  * the 1-NN graph: the reference builds the dense N x N sklearn distance matrix and argmins every row for
    N <= 70 000, and needs pyflann's approximate kd-tree above that (finch.py:26-37).  Here it is the fused
    similarity-GEMM + top-k kernel with k = 1 and the diagonal masked (csrc/topk.hip) — exact at any N.
  * `orig_dist` is never dense: the only uses of it are `orig_dist * adj` at the adjacency's non-zeros
    (finch.py:52,149), so it is a sparse matrix with the cosine distances of exactly those pairs (rows gathered on
    the device, `slic_pair_distance`).
  * cluster means (`cool_mean`) use the k-means M-step kernel (per-cluster sums in ascending row order) instead of a
    host cumsum.
Graph algebra (A + I)(A + I)^T and connected components stay scipy.sparse on the host, as in the reference.
Only distance='cosine' (what cluster_masks.py:81 passes) is supported.
"""
import numpy as np
import scipy.sparse as sp
import torch

from .. import _lib
from .._lib import call, ptr, stream
from ..evaluate import cosine_topk
from .kmeans_hip import HipKernels


def _dev(mat):
    if torch.is_tensor(mat):
        return mat.detach().to(device="cuda", dtype=torch.float32).contiguous()
    return torch.as_tensor(np.ascontiguousarray(mat, dtype=np.float32)).cuda()


def _pair_cosine_distance(mat_d, rows, cols, chunk=1 << 20):
    """1 - cos(mat[rows[i]], mat[cols[i]]) for index arrays on the host; rows gathered on the device in chunks"""
    out = np.empty(len(rows), np.float32)
    D = mat_d.shape[1]
    for s in range(0, len(rows), chunk):
        r = torch.from_numpy(np.ascontiguousarray(rows[s:s + chunk]).astype(np.int64)).cuda()
        c = torch.from_numpy(np.ascontiguousarray(cols[s:s + chunk]).astype(np.int64)).cuda()
        x, y = mat_d.index_select(0, r), mat_d.index_select(0, c)
        d = torch.empty(len(r), dtype=torch.float32, device=mat_d.device)
        call("slic_pair_distance", ptr(x), ptr(y), len(r), D, 0, ptr(d), stream())
        out[s:s + chunk] = d.cpu().numpy()
    return out


def clust_rank(mat, initial_rank=None, distance='cosine'):
    """returns (A, orig_dist): A = (first-neighbour adjacency + I)(...)^T with zero diagonal (lil), orig_dist = sparse
    matrix of the cosine distances at A's non-zeros (or [] when initial_rank was given, like the reference)"""
    if distance != 'cosine':
        raise NotImplementedError("FINCH on the GPU supports distance='cosine' (what SLIC passes, cluster_masks.py:81)")
    s = mat.shape[0]
    mat_d = _dev(mat)
    if initial_rank is not None:
        orig_dist = []
    else:
        idx, _ = cosine_topk(mat_d, None, k=1)                     # exact 1-NN, self excluded (np.fill_diagonal(.., 1e12))
        initial_rank = idx.view(-1).cpu().numpy().astype(np.int64)
        orig_dist = None
    A = sp.csr_matrix((np.ones_like(initial_rank, dtype=np.float32), (np.arange(0, s), initial_rank)), shape=(s, s))
    A = A + sp.eye(s, dtype=np.float32, format='csr')
    A = A @ A.T
    A = A.tolil()
    A.setdiag(0)
    if orig_dist is None:
        Ac = A.tocoo()
        keep = Ac.data != 0
        rows, cols = Ac.row[keep], Ac.col[keep]
        d = _pair_cosine_distance(mat_d, rows, cols)
        orig_dist = sp.csr_matrix((d, (rows, cols)), shape=(s, s))
    return A, orig_dist


def get_clust(a, orig_dist, min_sim=None):
    if min_sim is not None:
        # a[np.where((orig_dist * a.toarray()) > min_sim)] = 0, on the sparse pattern
        w = sp.csr_matrix(orig_dist).multiply(sp.csr_matrix(a)).tocoo()
        cut = w.data > min_sim
        if cut.any():
            a = sp.lil_matrix(a)
            a[w.row[cut], w.col[cut]] = 0
    num_clust, u = sp.csgraph.connected_components(csgraph=a, directed=True, connection='weak', return_labels=True)
    return u, num_clust


def cool_mean(M, u):
    """mean of the rows of M per label in u (labels 0..n-1 as connected_components / np.unique produce them)"""
    u = np.asarray(u)
    K = int(u.max()) + 1
    Md = _dev(M)
    N, D = Md.shape
    Dp = (D + 3) // 4 * 4
    if Dp != D:
        P = torch.zeros(N, Dp, dtype=torch.float32, device=Md.device)
        P[:, :D] = Md
        Md = P
    sums = torch.empty(K * Dp, dtype=torch.float32, device=Md.device)
    counts = torch.empty(K, dtype=torch.float32, device=Md.device)
    HipKernels().accumulate(Md, torch.from_numpy(u.astype(np.int32)).cuda(), K, sums, counts)
    out = (sums.view(K, Dp) / counts[:, None])[:, :D]
    return out.cpu().numpy()


def get_merge(c, u, data):
    if len(c) != 0:
        _, ig = np.unique(c, return_inverse=True)
        c = u[ig]
    else:
        c = u
    mat = cool_mean(data, c)
    return c, mat


def update_adj(adj, d):
    """keep one merge at a time: the two closest linked pairs (finch.py:85-94); d is the sparse distance matrix"""
    adj = sp.coo_matrix(adj)
    nz = adj.data != 0
    rows, cols = adj.row[nz], adj.col[nz]
    dv = np.asarray(sp.csr_matrix(d)[rows, cols]).reshape(-1)
    v = np.argsort(dv)[:2]
    a = sp.lil_matrix(adj.shape)
    a[[rows[v[0]], rows[v[1]]], [cols[v[0]], cols[v[1]]]] = 1
    return a


def req_numclust(c, data, req_clust, distance):
    iter_ = len(np.unique(c)) - req_clust
    c_, mat = get_merge([], c, data)
    for i in range(iter_):
        adj, orig_dist = clust_rank(mat, initial_rank=None, distance=distance)
        adj = update_adj(adj, orig_dist)
        u, _ = get_clust(adj, [], min_sim=None)
        c_, mat = get_merge(c_, u, data)
    return c_


def FINCH(data, initial_rank=None, req_clust=None, distance='cosine', ensure_early_exit=True, verbose=True):
    """same contract as the reference's FINCH: c [N, P] labels per partition, num_clust list, req_c or None"""
    _lib.load()
    if not torch.cuda.is_available():
        raise _lib.SlicError("FINCH needs a gfx950 device for its first-neighbour search (no CPU fallback)")
    if torch.is_tensor(data):
        data = data.detach().cpu().numpy()
    data = data.astype(np.float32)
    min_sim = None
    adj, orig_dist = clust_rank(data, initial_rank, distance)
    initial_rank = None
    group, num_clust = get_clust(adj, [], min_sim)
    c, mat = get_merge([], group, data)
    if verbose:
        print('Partition 0: {} clusters'.format(num_clust))
    if ensure_early_exit:
        if not isinstance(orig_dist, list):
            w = sp.csr_matrix(orig_dist).multiply(sp.csr_matrix(adj))
            min_sim = float(w.max()) if w.nnz else 0.0               # np.max(orig_dist * adj.toarray())
    exit_clust = 2
    c_ = c
    k = 1
    num_clust = [num_clust]
    while exit_clust > 1:
        adj, orig_dist = clust_rank(mat, initial_rank, distance)
        u, num_clust_curr = get_clust(adj, orig_dist, min_sim)
        c_, mat = get_merge(c_, u, data)
        num_clust.append(num_clust_curr)
        c = np.column_stack((c, c_))
        exit_clust = num_clust[-2] - num_clust_curr
        if num_clust_curr == 1 or exit_clust < 1:
            num_clust = num_clust[:-1]
            c = c[:, :-1]
            break
        if verbose:
            print('Partition {}: {} clusters'.format(k, num_clust[k]))
        k += 1
    if req_clust is not None:
        if req_clust not in num_clust:
            ind = [i for i, v in enumerate(num_clust) if v >= req_clust]
            req_c = req_numclust(c[:, ind[-1]], data, req_clust, distance)
        else:
            req_c = c[:, num_clust.index(req_clust)]
    else:
        req_c = None
    return c, num_clust, req_c
