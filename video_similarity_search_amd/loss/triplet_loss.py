"""
Drop-in for the reference's loss/triplet_loss.py on the SLIC training path:
    OnlineTripletLoss(margin, dist_metric)(embeddings[2B, D], labels[2B], sampling_strategy)  <- :86-116
    pdist(vectors, eps, dist_metric)                                                           <- :429-437
`sampling_strategy='noise_contrastive'` (NT-Xent over the 2B x 2B cosine matrix, T = 0.5, diagonal masked to
0, target (B + i) mod 2B) is one fused HIP forward kernel + one backward kernel (csrc/loss.hip) instead of
2B F.cosine_similarity launches + masked_fill + a Python target loop + F.cross_entropy.
The margin-triplet strategies `random_negative`, `random_semi_hard` (the shipped default,
config/custom_configs/*.yaml:13) and `fixed_semi_hard` (:205-227 with NegativeTripletSelector :230-360) run on the
device too: one pdist kernel, one selection kernel (one wave per anchor/positive pair, csrc/loss.hip
triplet_select_kernel) instead of the reference's per-pair Python loop with `random.choice` / `torch.where`, and the
fused margin-ranking-on-cosine-distances kernel for the loss.  Python's `random` is replaced by the device RNG (same
distribution: uniform over the same candidate set; `fixed_semi_hard` is deterministic and matches the reference
exactly).  `all_semi_hard` and the MemTripletLoss queue variants are not used by the shipped configs and raise.
"""
import itertools

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .._lib import call, ptr, stream

NCE_TEMPERATURE = 0.5      # hard-coded in the reference (loss/triplet_loss.py:99)


class _NTXent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, temperature):
        lib = _lib.load()
        if not emb.is_cuda:
            raise _lib.SlicError("noise_contrastive loss needs device embeddings (no CPU fallback)")
        e = emb.contiguous().float()
        n, D = e.shape
        ws = torch.empty(lib.slic_ntxent_workspace_bytes(n, D), dtype=torch.uint8, device=e.device)
        loss = torch.empty((), dtype=torch.float32, device=e.device)
        call("slic_ntxent_fwd", ptr(e), n, D, e.stride(0), float(temperature), ptr(loss), ptr(ws), stream())
        ctx.ws, ctx.n, ctx.D, ctx.T = ws, n, D, float(temperature)
        return loss

    @staticmethod
    def backward(ctx, g):
        dE = torch.empty(ctx.n, ctx.D, dtype=torch.float32, device=g.device)
        g = g.contiguous().float()
        call("slic_ntxent_bwd", ptr(ctx.ws), ctx.n, ctx.D, ctx.T, ptr(g), ptr(dE), ctx.D, stream())
        return dE, None


def ntxent_loss(embeddings, temperature=NCE_TEMPERATURE):
    D = embeddings.shape[1]
    if D % 2:
        embeddings = torch.nn.functional.pad(embeddings, (0, 1))
    return _NTXent.apply(embeddings, temperature)


class _MarginCos(torch.autograd.Function):
    """mean(max(0, (1 - cos(x, y)) - (1 - cos(x, z)) + margin)) — relu(ap_dists - an_dists + margin).mean() on cosine
    distances (loss/triplet_loss.py:216-227) and MarginRankingLoss(margin)(d_xy, d_xz, -1) (online_train.py:321-332)"""

    @staticmethod
    def forward(ctx, x, y, z, margin):
        x, y, z = (t.contiguous().float() for t in (x, y, z))
        n, D = x.shape
        state = torch.empty(n, 8, dtype=torch.float32, device=x.device)
        rowloss = torch.empty(n, dtype=torch.float32, device=x.device)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        call("slic_margin_cos_fwd", ptr(x), ptr(y), ptr(z), n, D, float(margin), ptr(state), ptr(rowloss), ptr(loss), stream())
        ctx.save_for_backward(x, y, z, state)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, y, z, state = ctx.saved_tensors
        n, D = x.shape
        dx, dy, dz = torch.empty_like(x), torch.empty_like(y), torch.empty_like(z)
        call("slic_margin_cos_bwd", ptr(x), ptr(y), ptr(z), ptr(state), n, D, ptr(g.contiguous().float()), ptr(dx), ptr(dy),
             ptr(dz), stream())
        return dx, dy, dz, None


def margin_cosine_loss(anchor, near, far, margin):
    """`near` should end up closer to `anchor` than `far` by `margin` in cosine distance (mean hinge)"""
    if not anchor.is_cuda:
        raise _lib.SlicError("margin_cosine_loss needs device tensors (no CPU fallback)")
    return _MarginCos.apply(anchor, near, far, margin)


_SELECT_MODES = {"random_negative": 0, "random_semi_hard": 1, "fixed_semi_hard": 2}


def get_triplets(embeddings, labels, margin, sampling_strategy, dist_metric='cosine'):
    """NegativeTripletSelector.get_triplets (loss/triplet_loss.py:275-309): [anchor idx], [positive idx], [negative idx]
    as int64 device tensors.  Pairs = combinations of the rows of every label with >= 2 rows, labels in ascending order
    (torch.unique), rows ascending — the reference's enumeration order."""
    if dist_metric != 'cosine':
        raise NotImplementedError("triplet mining with euclidean distances is not used by any SLIC config")
    lab_h = labels.detach().cpu().numpy()
    assert -1 not in lab_h                           # the reference's assert (:285)
    n = len(lab_h)
    anc, pos = [], []
    for lab in np.unique(lab_h):
        idx = np.nonzero(lab_h == lab)[0]
        if len(idx) < 2 or len(idx) == n:
            continue
        for a, p in itertools.combinations(idx.tolist(), 2):
            anc.append(a)
            pos.append(p)
    dev = embeddings.device
    if not anc:
        e = torch.empty(0, dtype=torch.long, device=dev)
        return e, e, e
    P = len(anc)
    D = pdist(embeddings.detach(), eps=0, dist_metric=dist_metric)
    anc_d = torch.tensor(anc, dtype=torch.int32, device=dev)
    pos_d = torch.tensor(pos, dtype=torch.int32, device=dev)
    neg_d = torch.empty(P, dtype=torch.int32, device=dev)
    u = torch.rand(P, dtype=torch.float32, device=dev)
    lab_d = labels.detach().to(device=dev, dtype=torch.int64).contiguous()
    call("slic_triplet_select", ptr(D), ptr(lab_d), n, ptr(anc_d), ptr(pos_d), P, float(margin),
         _SELECT_MODES[sampling_strategy], ptr(u), ptr(neg_d), stream())
    return anc_d.long(), pos_d.long(), neg_d.long()


class OnlineTripletLoss(nn.Module):
    def __init__(self, margin, dist_metric='cosine'):
        super(OnlineTripletLoss, self).__init__()
        self.margin = margin
        self.triplet_selector = None
        self.dist_metric = dist_metric

    # embeddings: [(batch_size * 2), dim_embedding] = cat(anchors, positives); labels: [(batch_size * 2)]
    def forward(self, embeddings, labels, sampling_strategy="random_negative"):
        if sampling_strategy == 'noise_contrastive':
            if self.dist_metric != 'cosine':
                raise NotImplementedError("noise_contrastive with euclidean pdist is not used by any SLIC config")
            return ntxent_loss(embeddings), 0
        if sampling_strategy in _SELECT_MODES:
            a, p, n = get_triplets(embeddings, labels, self.margin, sampling_strategy, self.dist_metric)
            if a.numel() == 0:
                return torch.zeros(1, requires_grad=True).mean(), 0          # loss/triplet_loss.py:222-223
            e = embeddings.float()
            # relu(ap_dists - an_dists + margin).mean() on 1 - cos (:216-227); index_select's backward scatters the grads
            loss = margin_cosine_loss(e.index_select(0, a), e.index_select(0, p), e.index_select(0, n), self.margin)
            return loss, int(a.numel())
        raise NotImplementedError(
            f"sampling_strategy={sampling_strategy!r}: not used by the shipped SLIC configs "
            "(all_semi_hard / adapted_hard are MemTripletLoss / ablation variants)")


def pdist(vectors, eps, dist_metric):
    """cosine (1 - cos) or euclidean distance matrix between all rows (loss/triplet_loss.py:429-437), one kernel"""
    v = vectors.contiguous().float()
    n, D = v.shape
    out = torch.empty(n, n, dtype=torch.float32, device=v.device)
    call("slic_pdist", ptr(v), n, D, float(eps), int(dist_metric == 'euclidean'), ptr(out), stream())
    return out
