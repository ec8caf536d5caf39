"""
First-neighbour hierarchical clustering (the FINCH algorithm: Sarfraz, Sharma, Stiefelhagen, "Efficient Parameter-free
Clustering Using First Neighbor Relations", CVPR 2019) on the MI355X — the clustering method every shipped SLIC config
selects (ITERCLUSTER.METHOD: finch).

Drop-in for the one call the reference makes (clustering/cluster_masks.py:79-86):
    FINCH(data, initial_rank=None, req_clust=None, distance='cosine', ensure_early_exit=True, verbose=True)
        -> (c [N, P] labels per partition, num_clust [P], req_c or None)
with the behaviour of the reference's clustering/finch.py:108-178 (pinned by tests/golden/finch.npz, produced by
importing that file).  The algorithm is the published one; the reference's file is third-party code distributed
under a research-only notice (clustering/finch.py:122-128) and nothing of it is reproduced here — this module is
written against the paper's definition and the golden outputs:

  level 0   every point i has a first neighbour k(i) (nearest other point, cosine).  Points i, j are linked when
            j = k(i), i = k(j) or k(i) = k(j); the partition is the connected components of the link graph.
  level l   the clusters of level l-1 are replaced by the means of their ORIGINAL rows and linked the same way; with
            `ensure_early_exit` a link whose weighted cosine distance (x 2 for a mutual first-neighbour pair, the
            reference's adjacency weight) exceeds the largest weighted link distance of level 0 is dropped.
  stop      when one cluster is left or the number of clusters stops decreasing.
  req_clust from the finest partition with at least req_clust clusters, merge the single closest linked pair per step.

How it is laid out for the GPU (and what differs from a host implementation):
  * the rows stay resident in HBM for the whole hierarchy (`_Hierarchy.rows`); each level's representatives are
    computed there (per-cluster sums in ascending row order on the k-means M-step kernel) and never visit the host;
  * first neighbours come from the fused similarity-GEMM + top-k kernel with k = 1 and the diagonal masked
    (csrc/topk.hip): exact at any N — a dense N x N distance matrix (or an approximate kd-tree beyond 70 000 rows,
    which is what the reference falls back to) is never built;
  * link distances are evaluated only for the linked pairs (`slic_pair_distance` on gathered rows);
  * the link graph is an explicit pair list (direct links + co-neighbour groups), not a sparse matrix product.
Connected components run on the host (scipy.sparse.csgraph), O(N) work on an O(N)-edge graph.
Only distance='cosine' (what cluster_masks.py:81 passes) is supported.
"""
import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import connected_components
import torch

from .. import _lib
from .._lib import call, ptr, stream
from ..evaluate import cosine_topk
from .kmeans_hip import HipKernels


class HipFinchKernels:
    """the device side of FINCH: resident rows, first neighbours (top-k kernel, k = 1, diagonal masked), link distances on
    gathered rows, per-cluster means (k-means M-step kernel).  `FINCH(..., kernels=)` takes another provider with the same
    four methods (tests of the host logic on a GPU-less machine pass a NumPy one as an ARGUMENT; the product has no other)."""

    def __init__(self):
        _lib.load()
        if not torch.cuda.is_available():
            raise _lib.SlicError("FINCH needs a gfx950 device for its first-neighbour search (no CPU fallback)")
        self.kern = HipKernels()

    def resident(self, mat):
        """fp32 device copy of the rows (no copy when they already are one)"""
        if torch.is_tensor(mat):
            return mat.detach().to(device="cuda", dtype=torch.float32).contiguous()
        return torch.as_tensor(np.ascontiguousarray(mat, dtype=np.float32)).cuda()

    def first_neighbours(self, rows):
        """index of the nearest OTHER row (cosine) for every row of a device matrix; a single row is its own neighbour"""
        n = rows.shape[0]
        if n == 1:
            return np.zeros(1, np.int64)
        idx, _ = cosine_topk(rows, None, k=1)
        return idx.view(-1).cpu().numpy().astype(np.int64)

    def pair_cosine_distance(self, rows, a, b, chunk=1 << 20):
        """1 - cos(rows[a[i]], rows[b[i]]) for host index arrays; the rows are gathered on the device in chunks"""
        out = np.empty(len(a), np.float32)
        D = rows.shape[1]
        for s in range(0, len(a), chunk):
            ia = torch.from_numpy(np.ascontiguousarray(a[s:s + chunk])).cuda()
            ib = torch.from_numpy(np.ascontiguousarray(b[s:s + chunk])).cuda()
            x, y = rows.index_select(0, ia), rows.index_select(0, ib)
            d = torch.empty(len(ia), dtype=torch.float32, device=rows.device)
            call("slic_pair_distance", ptr(x), ptr(y), len(ia), D, 0, ptr(d), stream())
            out[s:s + chunk] = d.cpu().numpy()
        return out

    def cluster_means(self, rows, assign, K):
        """per-cluster means of the rows on the device (sums in ascending row order: deterministic)"""
        N, D = rows.shape
        Dp = (D + 3) // 4 * 4                               # the M-step kernel wants 16-byte rows
        if Dp != D:
            key = (rows.data_ptr(), N, D)
            if getattr(self, "_pad_key", None) != key:
                padded = torch.zeros(N, Dp, dtype=torch.float32, device=rows.device)
                padded[:, :D] = rows
                self._pad_key, self._padded = key, padded
            acc = self._padded
        else:
            acc = rows
        sums = torch.empty(K * Dp, dtype=torch.float32, device=rows.device)
        counts = torch.empty(K, dtype=torch.float32, device=rows.device)
        lab = torch.from_numpy(assign.astype(np.int32)).to(rows.device)
        self.kern.accumulate(acc, lab, K, sums, counts)
        return (sums.view(K, Dp) / counts[:, None])[:, :D].contiguous()


def first_neighbours(rows):
    """module-level form of HipFinchKernels.first_neighbours (used by callers that only want the 1-NN map)"""
    return HipFinchKernels().first_neighbours(rows)


def link_pairs(nn):
    """the undirected link set of a first-neighbour map as two index arrays (a < b, each pair once):
    direct links {i, nn[i]} and every pair of points that share a first neighbour"""
    n = len(nn)
    i = np.arange(n, dtype=np.int64)
    lo, hi = np.minimum(i, nn), np.maximum(i, nn)
    # co-neighbour groups: sort the points by their neighbour; a member at position p of its group pairs with the p members before it
    order = np.argsort(nn, kind="stable")
    key = nn[order]
    start = np.flatnonzero(np.r_[True, key[1:] != key[:-1]])
    size = np.diff(np.r_[start, n])
    gstart = np.repeat(start, size)
    pos = np.arange(n) - gstart
    total = int(pos.sum())
    if total:
        second = np.repeat(order, pos)
        within = np.arange(total) - np.repeat(np.cumsum(pos) - pos, pos)
        first = order[np.repeat(gstart, pos) + within]
        lo = np.concatenate([lo, np.minimum(first, second)])
        hi = np.concatenate([hi, np.maximum(first, second)])
    keep = lo != hi
    code = np.unique(lo[keep] * n + hi[keep])
    return code // n, code % n


def components(n, a, b):
    """labels 0..C-1 of the connected components of an undirected pair list, numbered by their smallest member"""
    g = sp.coo_matrix((np.ones(len(a), np.int8), (a, b)), shape=(n, n))
    count, labels = connected_components(g, directed=False)
    return labels.astype(np.int64), int(count)


class _Hierarchy:
    """state of one FINCH run: the resident rows, the current assignment of every original row, the current
    representatives (device), and the partitions accepted so far"""

    def __init__(self, data, kernels):
        self.k = kernels
        self.rows = kernels.resident(data)                  # [N, D], stays in HBM
        self.N, self.D = self.rows.shape
        self.assign = None                                  # np.int64 [N]: cluster of every original row at the current level
        self.reps = self.rows                               # what the next level links: rows, then cluster means
        self.cut = None                                     # early-exit bound on link distances
        self.levels, self.sizes = [], []

    def means_of(self, assign, K):
        return self.k.cluster_means(self.rows, assign, K)

    def link_level(self, nn=None, want_max=False):
        """link the current representatives; returns (labels of the representatives, count, largest kept link distance)"""
        reps = self.reps
        n = reps.shape[0]
        if nn is None:
            nn = self.k.first_neighbours(reps)
        nn = np.asarray(nn, dtype=np.int64)
        a, b = link_pairs(nn)
        dmax = None
        if (self.cut is not None or want_max) and len(a):
            # The reference bounds and cuts on (distance x adjacency weight) (clustering/finch.py:44-45,144): its adjacency
            # (A+I)(A+I)^T counts a MUTUAL first-neighbour pair twice, so such a link stands at 2 d in both the level-0
            # bound and the later cuts; every other link (one-directional, or a shared first neighbour) at d.
            d = self.k.pair_cosine_distance(reps, a, b) * np.where((nn[a] == b) & (nn[b] == a), np.float32(2), np.float32(1))
            if self.cut is not None:
                keep = ~(d > self.cut)
                a, b, d = a[keep], b[keep], d[keep]
            dmax = float(d.max()) if len(d) else 0.0
        labels, count = components(n, a, b)
        return labels, count, dmax

    def descend(self, labels, count):
        """make `labels` (over the current representatives) the new level"""
        self.assign = labels if self.assign is None else labels[self.assign]
        self.reps = self.means_of(self.assign, count)

    def merge_closest_pair(self):
        """one agglomeration step of the req_clust refinement: of all first-neighbour links keep the single closest"""
        n = self.reps.shape[0]
        a, b = link_pairs(self.k.first_neighbours(self.reps))
        d = self.k.pair_cosine_distance(self.reps, a, b)
        j = int(np.lexsort((b, a, d))[0])                   # smallest distance; ties -> lowest pair
        labels, count = components(n, a[j:j + 1], b[j:j + 1])
        self.descend(labels, count)
        return count


def FINCH(data, initial_rank=None, req_clust=None, distance='cosine', ensure_early_exit=True, verbose=True, kernels=None):
    """Same call contract as the reference (clustering/finch.py:108): data [N, D] (ndarray or tensor; a device tensor is used
    in place), optional precomputed first neighbours `initial_rank` [N], optional `req_clust`.
    Returns (c, num_clust, req_c): c int [N, P] — column p = labels of partition p —, num_clust = clusters per
    partition, req_c = labels of the exactly-req_clust partition or None."""
    if distance != 'cosine':
        raise NotImplementedError("FINCH on the GPU supports distance='cosine' (what SLIC passes, cluster_masks.py:81)")
    kernels = HipFinchKernels() if kernels is None else kernels      # raises SlicError without libslic_hip.so / a gfx950 device
    h = _Hierarchy(data, kernels)
    # level 0: links between the points themselves; its largest link distance bounds the later levels' links
    labels, count, dmax = h.link_level(nn=initial_rank, want_max=ensure_early_exit and initial_rank is None)
    h.descend(labels, count)
    h.levels.append(h.assign.copy())
    h.sizes.append(count)
    if verbose:
        print('Partition 0: {} clusters'.format(count))
    if ensure_early_exit and dmax is not None:
        h.cut = dmax
    # coarser levels: stop at one cluster, or when a level removes no cluster; a level that removes exactly one is the last
    while h.sizes[-1] > 1:
        labels, count, _ = h.link_level()
        removed = h.sizes[-1] - count
        if count == 1 or removed < 1:
            break
        h.descend(labels, count)
        h.levels.append(h.assign.copy())
        h.sizes.append(count)
        if verbose:
            print('Partition {}: {} clusters'.format(len(h.sizes) - 1, count))
        if removed == 1:
            break
    c = np.stack(h.levels, axis=1)
    num_clust = list(h.sizes)
    req_c = None
    if req_clust is not None:
        if req_clust in num_clust:
            req_c = c[:, num_clust.index(req_clust)]
        else:
            finer = [p for p, v in enumerate(num_clust) if v >= req_clust]
            if not finer:
                raise ValueError("req_clust = {} exceeds the finest partition ({} clusters)".format(req_clust, num_clust[0]))
            r = _Hierarchy(h.rows, kernels)
            r.descend(h.levels[finer[-1]], num_clust[finer[-1]])
            count = num_clust[finer[-1]]
            while count > req_clust:
                count = r.merge_closest_pair()
            req_c = r.assign
    return c, num_clust, req_c
