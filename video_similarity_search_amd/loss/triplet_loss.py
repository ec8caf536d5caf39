"""
Drop-in for the InfoNCE branch of the reference's loss/triplet_loss.py:
    OnlineTripletLoss(margin, dist_metric)(embeddings[2B, D], labels[2B], sampling_strategy)  <- :86-116
    pdist(vectors, eps, dist_metric)                                                           <- :429-437
`sampling_strategy='noise_contrastive'` (NT-Xent over the 2B x 2B cosine matrix, T = 0.5, diagonal masked to
0, target (B + i) mod 2B) is one fused HIP forward kernel + one backward kernel (csrc/loss.hip) instead of
2B F.cosine_similarity launches + masked_fill + a Python target loop + F.cross_entropy.
The margin-triplet / semi-hard mining branches (:118-227) are host-bound Python in the reference and a
"next" row in SURVEY.md §8f; they raise here.
"""
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
        raise NotImplementedError(
            f"sampling_strategy={sampling_strategy!r}: triplet mining (loss/triplet_loss.py:118-360) is host-side "
            "Python in the reference and a 'next' row of the hot-path scope (SURVEY.md §8f)")


def pdist(vectors, eps, dist_metric):
    """cosine (1 - cos) or euclidean distance matrix between all rows (loss/triplet_loss.py:429-437), one kernel"""
    v = vectors.contiguous().float()
    n, D = v.shape
    out = torch.empty(n, n, dtype=torch.float32, device=v.device)
    call("slic_pdist", ptr(v), n, D, float(eps), int(dist_metric == 'euclidean'), ptr(out), stream())
    return out
