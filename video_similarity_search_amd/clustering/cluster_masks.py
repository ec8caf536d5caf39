"""
Drop-in for the reference's clustering/cluster_masks.py on the k-means path:
    preprocess_features_kmeans(data)                                   <- cluster_masks.py:30-34
    fit_cluster(embeddings, method, k, l2normalize, finch_partition)   <- cluster_masks.py:38-98
Same names, argument meaning, prints and return type (np.ndarray[N] labels).  method='kmeans'
 and method='finch' (§8f row 1: the method the shipped configs select) run on
the GPU; the other methods the reference dispatches to sklearn on the host raise.
"""
import numpy as np
import torch

from .. import _lib
from .._lib import call, ptr, stream
from .kmeans_hip import KMeans

_METHODS = ['DBSCAN', 'Agglomerative', 'OPTICS', 'kmeans', 'spherical_kmeans', 'finch']


def _to_device(embeddings):
    if not torch.cuda.is_available():
        raise _lib.SlicError("fit_cluster needs a gfx950 device (no CPU fallback)")
    if not torch.is_tensor(embeddings):
        embeddings = torch.as_tensor(np.ascontiguousarray(embeddings, dtype=np.float32))
    return embeddings.detach().to(device="cuda", dtype=torch.float32).contiguous()


def preprocess_features_kmeans(data, kernels=None):
    """row L2-normalise: data / torch.norm(data, dim=1, keepdim=True) (no epsilon), on the device.
    Returns a tensor on the device the data ended up on (CPU input is moved to the current GPU)."""
    if kernels is not None:                 # explicit kernel provider (see KMeans(kernels=...)): tests of the host logic
        x = kernels.to_device(data)
        out = torch.empty_like(x)
        kernels.l2norm_rows(x, out)
    else:
        x = _to_device(data)
        out = torch.empty_like(x)
        N, D = x.shape
        call("slic_l2norm_rows", ptr(x), N, D, x.stride(0), ptr(out), out.stride(0), stream())
    print('l2-normalized data')
    return out


def fit_cluster(embeddings, method='Agglomerative', k=1000, l2normalize=True, finch_partition=0,
                n_init=10, init='k-means++', process_group=None, random_state=None, kernels=None, exchange=None):
    """Reference signature + keyword-only extras (n_init / init / process_group / random_state / kernels / exchange) that default
    to the reference's behaviour: KMeans(n_clusters=k, n_init=10).fit(embeddings).labels_.
    process_group: `embeddings` is this rank's row shard (rank order == row order), the returned labels are this rank's.
    exchange: the sharded Lloyd iteration's one collective — 'allreduce' (RCCL through torch.distributed; the default), 'allgather', or
    'oneshot' (the library's one-shot all-to-all over peer-mapped memory, csrc/oneshot.hip); None reads SLIC_KMEANS_EXCHANGE."""

    assert (method in _METHODS)
    print("Clustering with {}...".format(method))
    if method == 'finch':
        # cluster_masks.py:79-86: FINCH(embeddings, distance='cosine'), take partition `finch_partition`
        from .finch import FINCH
        c, num_clust, req_c = FINCH(embeddings, distance='cosine')      # rows go to (or stay on) the device once
        PARTITION = finch_partition
        labels = c[:, PARTITION]
        n_clusters = num_clust[PARTITION]
        print('Taking partition {} from finch'.format(PARTITION))
        print("Fitted " + str(n_clusters) + " clusters with " + str(method))
        return labels
    if method not in ('kmeans', 'spherical_kmeans'):
        raise NotImplementedError(
            f"method={method!r}: 'kmeans', 'spherical_kmeans' and 'finch' are on the MI355X hot path (SURVEY.md §8); the "
            "reference runs the others on the host through sklearn")
    x = _to_device(embeddings) if kernels is None else kernels.to_device(embeddings)
    if method == 'spherical_kmeans':
        # cluster_masks.py:73-77: SphericalKMeans(n_clusters=k).fit(embeddings) (spherecluster: normalises the rows itself,
        # n_init=10, k-means++, centres renormalised every iteration).  spherecluster is not vendored: parity unpinned.
        print('clustering with spherical kmeans with k={}'.format(k))
        print(tuple(x.shape))
        km = KMeans(n_clusters=k, n_init=n_init, init=init, process_group=process_group, random_state=random_state,
                    spherical=True, kernels=kernels, exchange=exchange).fit(x)
    else:
        print("k:", k)
        if l2normalize:
            x = preprocess_features_kmeans(x, kernels)
        km = KMeans(n_clusters=k, n_init=n_init, init=init, process_group=process_group,
                    random_state=random_state, kernels=kernels, exchange=exchange).fit(x)
    labels = km.labels_
    print(labels.shape)
    n_clusters = len(set(labels.tolist())) - (1 if -1 in labels else 0)
    print("Fitted " + str(n_clusters) + " clusters with " + str(method))
    fit_cluster.last_model = km
    return labels
