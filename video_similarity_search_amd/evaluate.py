"""
Drop-in for the retrieval / embedding-extraction functions of the reference's evaluate.py and
iic_retrieve_clips.py that sit on the hot path:

    get_distance_matrix(x, y=None, dist_metric)           <- evaluate.py:208-223
    get_closest_data_mat(distance_matrix, top_k)          <- evaluate.py:226-231
    get_topk_acc(distance_matrix, x_labels, y_labels, top_ks)   <- evaluate.py:287-307
    evaluate(model, data_loader, ...) / get_embeddings_and_labels(...)   <- evaluate.py:146-205, 310-350
    topk_retrieval(args | arrays)                          <- iic_retrieve_clips.py:275-314

The reference computes an N_q x N_g sklearn cosine_distances matrix on the host and argsorts every row.
Here the matrix (when a caller really wants it) comes from the fp32-MFMA gather-GEMM with a
(1 - s, clamp at 0) epilogue, and the top-k paths never build it: `cosine_topk` fuses the similarity GEMM
with a per-query streaming top-k (csrc/topk.hip).  Plot/heat-map helpers of evaluate.py are out of scope.
"""
import ctypes
import json
import os

import numpy as np
import torch

from . import _lib
from ._lib import SlicConvArgs, call, ptr, stream


def _dev(x):
    if not torch.cuda.is_available():
        raise _lib.SlicError("retrieval needs a gfx950 device (no CPU fallback)")
    if not torch.is_tensor(x):
        x = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32))   # float64 .npy features are cast (D7)
    return x.detach().to(device="cuda", dtype=torch.float32).contiguous()


def _normalize(x, Dp):
    """sklearn normalize(X) rows; feature dim zero-padded to Dp (a multiple of 8)"""
    N, D = x.shape
    out = torch.empty(N, D, dtype=torch.float32, device=x.device)
    call("slic_normalize_rows", ptr(x), N, D, x.stride(0), ptr(out), stream())
    if Dp != D:
        p = torch.zeros(N, Dp, dtype=torch.float32, device=x.device)
        p[:, :D] = out
        out = p
    return out


def cosine_topk_sharded(queries, gallery_shard, k, process_group, row_offset=None):
    """Gallery sharded by rows across the ranks of `process_group` (queries replicated): every rank searches its
    shard, the [Nq, k] lists are all-gathered (RCCL) and merged on every GPU.  Returned indices are GLOBAL gallery
    rows (rank order == row order unless `row_offset` is given)."""
    import torch.distributed as dist
    W = dist.get_world_size(process_group)
    g = _dev(gallery_shard)
    n_loc = torch.tensor([g.shape[0]], dtype=torch.int64, device=g.device)
    sizes = torch.empty(W, dtype=torch.int64, device=g.device)
    dist.all_gather_into_tensor(sizes, n_loc, group=process_group)
    if row_offset is None:
        row_offset = int(sizes[: dist.get_rank(process_group)].sum().item())
    kk = min(k, g.shape[0])
    idx, dst = cosine_topk(queries, g, k=kk)
    Nq = idx.shape[0]
    pi = torch.full((Nq, k), -1, dtype=torch.int32, device=g.device)
    pd = torch.full((Nq, k), float("inf"), dtype=torch.float32, device=g.device)
    pi[:, :kk] = idx + row_offset
    pd[:, :kk] = dst
    api = torch.empty(W * Nq * k, dtype=torch.int32, device=g.device)
    apd = torch.empty(W * Nq * k, dtype=torch.float32, device=g.device)
    dist.all_gather_into_tensor(api, pi.reshape(-1), group=process_group)
    dist.all_gather_into_tensor(apd, pd.reshape(-1), group=process_group)
    out_i = torch.empty(Nq, k, dtype=torch.int32, device=g.device)
    out_d = torch.empty(Nq, k, dtype=torch.float32, device=g.device)
    call("slic_topk_merge_lists", ptr(apd), ptr(api), W, Nq, k, ptr(out_i), ptr(out_d), stream())
    return out_i, out_d


def cosine_topk(queries, gallery=None, k=20):
    """indices [Nq, k] (int32) and cosine distances [Nq, k] of the k nearest gallery rows, ascending;
    gallery=None searches the queries themselves with the diagonal excluded (evaluate.py:221-222)."""
    lib = _lib.load()
    q = _dev(queries)
    self_mask = gallery is None
    g = q if self_mask else _dev(gallery)
    D = q.shape[1]
    Dp = (D + 7) // 8 * 8
    qn = _normalize(q, Dp)
    gn = qn if self_mask else _normalize(g, Dp)
    Nq, Ng = qn.shape[0], gn.shape[0]
    idx = torch.empty(Nq, k, dtype=torch.int32, device=q.device)
    dist = torch.empty(Nq, k, dtype=torch.float32, device=q.device)
    ws = _lib.workspace(lib.slic_cosine_topk_workspace_bytes(Nq, Ng, k), q.device, "topk")
    call("slic_cosine_topk", ptr(qn), Nq, ptr(gn), Ng, Dp, k, int(self_mask), ptr(idx), ptr(dist), ptr(ws), stream())
    return idx, dist


def get_distance_matrix(x_embeddings, y_embeddings=None, dist_metric='cosine'):
    """full distance matrix as np.ndarray, like the reference (validation-sized inputs; the top-k paths do not
    need it).  Self-distance diagonal = +inf when y is None."""
    assert (dist_metric in ['cosine', 'euclidean'])
    x = _dev(x_embeddings)
    y = x if y_embeddings is None else _dev(y_embeddings)
    Nx, Ny, D = x.shape[0], y.shape[0], x.shape[1]
    Np = (Ny + 3) // 4 * 4                  # the GEMM epilogue stores 16-byte chunks: row stride and N are multiples of 4
    if dist_metric == 'cosine':
        out = torch.empty(Nx, Np, dtype=torch.float32, device=x.device)[:, :Ny]
        # D = max(0, 1 - x_hat . y_hat): the gather-GEMM as a plain GEMM (1 tap) with scale -1, shift +1, ReLU
        Dp = (D + 31) // 32 * 32
        xn, yn = _normalize(x, Dp), (None if y_embeddings is None else _normalize(y, Dp))
        yn = xn if yn is None else yn
        tab = np.zeros((Dp // 4, 4), np.int32)
        tab[:, 0] = np.arange(Dp // 4) * 4
        tab[:, 1] = (1 << 3) | (1 << 10) | (1 << 17)          # tap offset (0, 0, 0)
        tab[:, 2] = np.arange(Dp // 4) * 4
        tab[:, 3] = 128 | (128 << 8) | (128 << 16)
        tabd = torch.from_numpy(tab).to(x.device)
        sc = torch.full((Np,), -1.0, device=x.device)
        sh = torch.full((Np,), 1.0, device=x.device)
        a = SlicConvArgs()
        a.src, a.wgt, a.dst, a.tab = xn.data_ptr(), yn.data_ptr(), out.data_ptr(), tabd.data_ptr()
        a.src_bytes, a.wgt_bytes = _lib.u32_bytes(xn, 'x_embeddings'), _lib.u32_bytes(yn, 'y_embeddings')
        a.scale, a.shift, a.relu = sc.data_ptr(), sh.data_ptr(), 1
        a.M, a.N, a.nchunks = Nx, Np, Dp // 4       # columns Ny..Np-1: rows past yn's range read as zeros
        a.Cs, a.Ts, a.Hs, a.Ws = Dp, 1, 1, 1
        a.Ga = a.Gb = a.Gc = 1
        a.sa = a.sb = a.sc = 1
        a.ldw, a.ldo = Dp, Np
        call("slic_conv_gemm", ctypes.byref(a), 0, stream())
    else:
        out = torch.empty(Nx, Ny, dtype=torch.float32, device=x.device)
        call("slic_pairwise_euclidean", ptr(x), Nx, ptr(y), Ny, D, ptr(out), stream())
    distance_matrix = out.cpu().numpy()
    if y_embeddings is None:
        np.fill_diagonal(distance_matrix, float('inf'))
    return distance_matrix


def get_closest_data_mat(distance_matrix, top_k):
    """top_k smallest per row, sorted (evaluate.py:226-231) — host numpy on an already materialised matrix"""
    idx = np.argpartition(distance_matrix, top_k, axis=-1)
    unsorted = np.take_along_axis(distance_matrix, idx[:, :top_k], axis=-1)
    order = np.argsort(unsorted, axis=-1)
    return np.take_along_axis(idx, order, axis=-1)


def _acc_from_indices(topk_indices, x_labels, y_labels, top_ks):
    x_labels = np.asarray(x_labels)
    y_labels = np.asarray(y_labels)
    lab = y_labels[topk_indices]                       # [Nq, kmax]
    hit = lab == x_labels[:, None]
    return np.array([hit[:, :k].any(axis=1).mean() for k in top_ks])


def get_topk_acc(distance_matrix, x_labels, y_labels=None, top_ks=[1, 5, 10, 20]):
    """evaluate.py:287-307 on a materialised matrix (kept for drop-in callers)"""
    topk_indices = get_closest_data_mat(distance_matrix, top_k=top_ks[-1])
    if y_labels is None:
        y_labels = x_labels
    return _acc_from_indices(topk_indices, x_labels, y_labels, top_ks)


def get_topk_acc_from_embeddings(x_embeddings, x_labels, y_embeddings=None, y_labels=None, top_ks=[1, 5, 10, 20]):
    """the same accuracies without the matrix: fused GEMM + top-k on the device"""
    idx, _ = cosine_topk(x_embeddings, y_embeddings, k=top_ks[-1])
    if y_labels is None:
        y_labels = x_labels
    return _acc_from_indices(idx.cpu().numpy(), x_labels, y_labels, top_ks)


def topk_retrieval(args=None, X_train=None, y_train=None, X_test=None, y_test=None, ks=(1, 5, 10, 20, 50)):
    """iic_retrieve_clips.py:275-314.  Either `args.feature_dir` holds {train,test}_{feature,class}.npy
    ([V, 10, D] features averaged over the 10 clips, :280,287) or arrays are passed directly.
    Returns {k: correct}; writes topk_correct.json next to the features like the reference."""
    feature_dir = getattr(args, "feature_dir", None) if args is not None else None
    if feature_dir is not None:
        X_train = np.load(os.path.join(feature_dir, 'train_feature.npy'))
        y_train = np.load(os.path.join(feature_dir, 'train_class.npy'))
        X_test = np.load(os.path.join(feature_dir, 'test_feature.npy'))
        y_test = np.load(os.path.join(feature_dir, 'test_class.npy'))
    X_train, X_test = np.asarray(X_train), np.asarray(X_test)
    y_train, y_test = np.asarray(y_train), np.asarray(y_test)
    if X_train.ndim == 3:
        X_train = np.mean(X_train, 1)
        y_train = y_train[:, 0]
    if X_test.ndim == 3:
        X_test = np.mean(X_test, 1)
        y_test = y_test[:, 0]
    X_train = X_train.reshape((-1, X_train.shape[-1]))
    X_test = X_test.reshape((-1, X_test.shape[-1]))
    y_train, y_test = y_train.reshape(-1), y_test.reshape(-1)
    ks = list(ks)
    idx, _ = cosine_topk(X_test, X_train, k=min(max(ks), len(X_train)))
    lab = y_train[idx.cpu().numpy()]
    hit = lab == y_test[:, None]
    topk_correct = {k: int(hit[:, :k].any(axis=1).sum()) for k in ks}
    for k in ks:
        correct, total = topk_correct[k], len(X_test)
        print('Top-{}, correct = {:.2f}, total = {}, acc = {:.3f}'.format(k, correct, total, correct / total))
    if feature_dir is not None:
        with open(os.path.join(feature_dir, 'topk_correct.json'), 'w') as fp:
            json.dump(topk_correct, fp)
    return topk_correct


# ------------------------------------------------------------------------------------------------
def evaluate(model, data_loader, device=None, is_master_proc=True, gather=True):
    """evaluate.py:146-205: eval-mode encoder over a loader of (input [b,3,T,S,S], targets [b], info, indexes [b]),
    no grad.  gather=True is the reference's return contract — per batch the (embedding, label, index) triples are
    all-gathered across ranks (misc/distributed_helper) and moved to the host: (Tensor[N',D] cpu, list[int], list[int]).
    gather=False keeps THIS rank's rows resident on its GPU: no collective, no per-batch D2H;
    returns (Tensor[n_local, D] on the device, list[int], list[int]) for the sharded k-means / retrieval to consume."""
    from .misc import distributed_helper as du_helper
    model.eval()
    embedding, vid_info, idxs = [], [], []
    world = du_helper.get_world_size()
    dev = torch.device("cuda") if device is None else torch.device(device)
    with torch.no_grad():
        for batch in data_loader:
            inp, targets, _info, indexes = batch
            inp = inp.to(dev, non_blocking=True)
            embedd = model(inp)
            if isinstance(embedd, tuple):
                embedd = embedd[0]
            embedd = embedd.flatten(1)
            if not gather:
                embedding.append(embedd.detach())
                vid_info.extend(torch.as_tensor(targets).tolist())
                idxs.extend(torch.as_tensor(indexes).tolist())
                continue
            targets = torch.as_tensor(targets).to(dev)
            indexes = torch.as_tensor(indexes).to(dev)
            if world > 1:
                embedd, targets, indexes = du_helper.all_gather([embedd, targets, indexes])
            embedding.append(embedd.detach().cpu())
            vid_info.extend(targets.cpu().tolist())
            idxs.extend(indexes.cpu().tolist())
    return torch.cat(embedding, dim=0), vid_info, idxs


def get_embeddings_and_labels(args, cfg, model, cuda, device, data_loader, split='val', is_master_proc=True,
                              load_pkl=False, save_pkl=False, gather=True):
    """evaluate.py:310-350 (same positional signature; the optional pkl cache uses torch.save/torch.load).
    gather=False: this rank's shard stays on its GPU (see evaluate); the pkl cache then holds per-rank files."""
    out_dir = getattr(cfg, "OUTPUT_PATH", None) if cfg is not None else None
    tag = split if gather else "{}_rank{}".format(split, torch.distributed.get_rank() if torch.distributed.is_initialized() else 0)
    names = [f"{tag}_embeddings.pkl", f"{tag}_labels.pkl", f"{tag}_idxs.pkl"]
    if load_pkl and out_dir and all(os.path.exists(os.path.join(out_dir, n)) for n in names):
        emb, labels, idxs = (torch.load(os.path.join(out_dir, n)) for n in names)
        return (emb if gather else emb.cuda()), labels, idxs
    embeddings, labels, idxs = evaluate(model, data_loader, device=device, is_master_proc=is_master_proc, gather=gather)
    if save_pkl and out_dir and (is_master_proc or not gather):
        for n, v in zip(names, (embeddings.cpu(), labels, idxs)):
            torch.save(v, os.path.join(out_dir, n))
    return embeddings, labels, idxs
