"""
Drop-in for the reference's models/triplet_net.py:7-34 — the validation-time wrapper
(online_train.py:521): three encoder passes and the anchor-positive / anchor-negative distances.
The encoder passes run on the HIP plan (models/resnet.py); the per-row distances are a fused HIP
kernel as well (csrc/loss.hip: slic_pair_distance).
"""
import torch
import torch.nn as nn

from .._lib import call, ptr, stream


class _PairDistance(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, euclid):
        x = x.contiguous().float()
        y = y.contiguous().float()
        out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        call("slic_pair_distance", ptr(x), ptr(y), x.shape[0], x.shape[1], int(euclid), ptr(out), stream())
        ctx.save_for_backward(x, y)
        ctx.euclid = int(euclid)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        g = g.contiguous().float()
        dx, dy = torch.empty_like(x), torch.empty_like(y)
        call("slic_pair_distance_bwd", ptr(x), ptr(y), ptr(g), x.shape[0], x.shape[1], ctx.euclid, ptr(dx), ptr(dy), stream())
        return dx, dy, None


def pair_distance(x, y, dist_metric):
    """rowwise 1 - cos(x, y) (per-norm clamp 1e-8, like F.cosine_similarity) or ||x - y + 1e-6||_2
    (F.pairwise_distance(x, y, 2): eps added to the difference); an ordinary autograd node, as in the reference"""
    return _PairDistance.apply(x, y, dist_metric == 'euclidean')


class Tripletnet(nn.Module):
    def __init__(self, embeddingnet, dist_metric='cosine'):
        super(Tripletnet, self).__init__()
        self.embeddingnet = embeddingnet
        assert dist_metric in ['cosine', 'euclidean']
        self.dist_metric = dist_metric

    def forward(self, x, y, z):
        embedded_x = self.embeddingnet(x)
        embedded_y = self.embeddingnet(y)
        embedded_z = self.embeddingnet(z)
        if isinstance(embedded_x, tuple):
            embedded_x = embedded_x[0]
        if isinstance(embedded_y, tuple):
            embedded_y = embedded_y[0]
        if isinstance(embedded_z, tuple):
            embedded_z = embedded_z[0]
        # validation.py:12-151 runs this under no_grad; when gradients are enabled the distances carry a graph like the
        # reference's F.cosine_similarity / F.pairwise_distance (models/triplet_net.py:28-32)
        dist_a = pair_distance(embedded_x, embedded_y, self.dist_metric)
        dist_b = pair_distance(embedded_x, embedded_z, self.dist_metric)
        return dist_a, dist_b, embedded_x, embedded_y, embedded_z
