"""
Drop-in for the reference's loss/triplet_loss.py on the SLIC training path:
    OnlineTripletLoss(margin, dist_metric)(embeddings[2B, D], labels[2B], sampling_strategy)  <- :86-116
    pdist(vectors, eps, dist_metric)                                                           <- :429-437
`sampling_strategy='noise_contrastive'` (NT-Xent over the 2B x 2B cosine matrix, T = 0.5, diagonal masked to
0, target (B + i) mod 2B) is one fused HIP forward kernel + one backward kernel (csrc/loss.hip) instead of
2B F.cosine_similarity launches + masked_fill + a Python target loop + F.cross_entropy.
`dist_metric='euclidean'` (LOSS.DIST_METRIC; no shipped config selects it) runs the same strategies on F.pairwise_distance:
the distance-matrix and selection kernels take the metric as a flag, the margin term and the noise-contrastive loss have
euclidean kernels of their own (margin_euclid_*, ntxent_euclid_*).
The margin-triplet strategies `random_negative`, `random_semi_hard` (the shipped default,
config/custom_configs/*.yaml:13) and `fixed_semi_hard` (:205-227 with NegativeTripletSelector :230-360) run on the
device too: one pdist kernel, one selection kernel (one wave per anchor/positive pair, csrc/loss.hip
triplet_select_kernel) instead of the reference's per-pair Python loop with `random.choice` / `torch.where`, and the
fused margin-ranking-on-cosine-distances kernel for the loss.  Python's `random` is replaced by the device RNG (same
distribution: uniform over the same candidate set; `fixed_semi_hard` is deterministic and matches the reference
exactly).
`all_semi_hard` (:118-203: InfoNCE of every anchor/positive pair against NUM_NEGATIVES = 5 picked negatives) and
`MemTripletLoss` (:9-81: the same margin-triplet loss against a 40-slot queue of past embeddings, default strategy
'adapted_hard') are built from the same pieces: distance matrix kernel, selection on the device, fused loss kernels.
Neither is selected by a shipped config; both restate the reference INCLUDING its quirks, which the goldens pin:
the hardest-easy fallback returns a position in the negatives list, `adapted_hard_sampling` returns nothing (so it
always falls back), and `all_semi_hard` draws its 5 negatives among the FIRST max(5, #semi-hard) rows of the
negatives list (it samples `enumerate` indices, not the candidates themselves).
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


class _MarginDist(torch.autograd.Function):
    """mean(max(0, d(x, y) - d(x, z) + margin)) — relu(ap_dists - an_dists + margin).mean() (loss/triplet_loss.py:212-227) and
    MarginRankingLoss(margin)(d_xy, d_xz, -1) (online_train.py:289-360) — with d = 1 - cos (F.cosine_similarity) or
    d = F.pairwise_distance (euclidean, eps 1e-6 inside the norm)"""

    @staticmethod
    def forward(ctx, x, y, z, margin, euclid):
        x, y, z = (t.contiguous().float() for t in (x, y, z))
        n, D = x.shape
        state = torch.empty(n, 8, dtype=torch.float32, device=x.device)
        rowloss = torch.empty(n, dtype=torch.float32, device=x.device)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        call("slic_margin_euclid_fwd" if euclid else "slic_margin_cos_fwd", ptr(x), ptr(y), ptr(z), n, D, float(margin), ptr(state),
             ptr(rowloss), ptr(loss), stream())
        ctx.save_for_backward(x, y, z, state)
        ctx.euclid = bool(euclid)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, y, z, state = ctx.saved_tensors
        n, D = x.shape
        dx, dy, dz = torch.empty_like(x), torch.empty_like(y), torch.empty_like(z)
        call("slic_margin_euclid_bwd" if ctx.euclid else "slic_margin_cos_bwd", ptr(x), ptr(y), ptr(z), ptr(state), n, D,
             ptr(g.contiguous().float()), ptr(dx), ptr(dy), ptr(dz), stream())
        return dx, dy, dz, None, None


def margin_distance_loss(anchor, near, far, margin, dist_metric='cosine'):
    """`near` should end up closer to `anchor` than `far` by `margin` in the metric's distance (mean hinge)"""
    assert dist_metric in ('cosine', 'euclidean')
    if not anchor.is_cuda:
        raise _lib.SlicError("margin_distance_loss needs device tensors (no CPU fallback)")
    return _MarginDist.apply(anchor, near, far, margin, dist_metric == 'euclidean')


def margin_cosine_loss(anchor, near, far, margin):
    return margin_distance_loss(anchor, near, far, margin, 'cosine')


class _NTXentEuclid(torch.autograd.Function):
    """'noise_contrastive' on euclidean distances (loss/triplet_loss.py:97-116 with dist_metric='euclidean'): the logits are
    (1 - ||e_i - e_j||) / T with the diagonal set to 0, the target of row i is (n / 2 + i) mod n"""

    @staticmethod
    def forward(ctx, emb, temperature):
        if not emb.is_cuda:
            raise _lib.SlicError("noise_contrastive loss needs device embeddings (no CPU fallback)")
        e = emb.contiguous().float()
        n, D = e.shape
        buf = torch.empty(2 * n * n + n + 1, dtype=torch.float32, device=e.device)
        dist, wm, rowloss, loss = buf[:n * n], buf[n * n:2 * n * n], buf[2 * n * n:2 * n * n + n], buf[-1]
        call("slic_ntxent_euclid_fwd", ptr(e), n, D, float(temperature), ptr(dist), ptr(wm), ptr(rowloss), ptr(loss), stream())
        ctx.save_for_backward(e, dist, wm)
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        e, dist, wm = ctx.saved_tensors
        n, D = e.shape
        dE = torch.empty_like(e)
        call("slic_ntxent_euclid_bwd", ptr(e), ptr(dist), ptr(wm), n, D, ptr(g.contiguous().float()), ptr(dE), stream())
        return dE, None


class _InfoNCERows(torch.autograd.Function):
    """mean_i -log(exp(cos(x_i, y_i0)/T) / sum_j exp(cos(x_i, y_ij)/T)) over gathered rows (x [P, D], y [P, NY, D])"""

    @staticmethod
    def forward(ctx, x, y, temperature):
        x, y = x.contiguous().float(), y.contiguous().float()
        P, NY, D = y.shape
        state = torch.empty(P, 18, dtype=torch.float32, device=x.device)
        rowloss = torch.empty(P, dtype=torch.float32, device=x.device)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        call("slic_infonce_rows_fwd", ptr(x), ptr(y), P, NY, D, float(temperature), ptr(state), ptr(rowloss), ptr(loss), stream())
        ctx.save_for_backward(x, y, state)
        ctx.T = float(temperature)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, y, state = ctx.saved_tensors
        P, NY, D = y.shape
        dx, dy = torch.empty_like(x), torch.empty_like(y)
        call("slic_infonce_rows_bwd", ptr(x), ptr(y), ptr(state), P, NY, D, ctx.T, ptr(g.contiguous().float()), ptr(dx), ptr(dy),
             stream())
        return dx, dy, None


_SELECT_MODES = {"random_negative": 0, "random_semi_hard": 1, "fixed_semi_hard": 2, "adapted_hard": 3}
NUM_NEGATIVES = 5          # hard-coded in the reference's all_semi_hard branch (loss/triplet_loss.py:120)


def _anchor_positive_pairs(lab_h):
    """combinations of the rows of every label with >= 2 rows (and at least one row of another label), labels in ascending
    order (torch.unique), rows ascending — the reference's enumeration order (loss/triplet_loss.py:130-152, 289-303)"""
    n = len(lab_h)
    anc, pos = [], []
    for lab in np.unique(lab_h):
        idx = np.nonzero(lab_h == lab)[0]
        if len(idx) < 2 or len(idx) == n:
            continue
        for a, p in itertools.combinations(idx.tolist(), 2):
            anc.append(a)
            pos.append(p)
    return anc, pos


def all_semi_hard_negatives(embeddings, labels, margin, anc, pos):
    """[P, NUM_NEGATIVES] row indices as the reference's all_semi_hard branch ends up using them (:158-183): with L = the number
    of negatives j with d(a,p) + margin - d(a,j) > 0, topped up to 5 with the closest ones, it draws 5 distinct ENUMERATE
    indices of that list — i.e. 5 distinct rows among the first max(L, 5) rows of the anchor's negatives list (ascending row
    order).  With L <= 5 that is exactly the first five negatives; beyond, a uniform 5-subset (device RNG here)."""
    dev = embeddings.device
    n = embeddings.shape[0]
    P = len(anc)
    Dm = pdist(embeddings.detach(), eps=0, dist_metric='cosine')
    anc_d = torch.tensor(anc, dtype=torch.int32, device=dev)
    pos_d = torch.tensor(pos, dtype=torch.int32, device=dev)
    lab_d = labels.detach().to(device=dev, dtype=torch.int64).contiguous()
    u = torch.rand((P, NUM_NEGATIVES), dtype=torch.float32, device=dev)
    neg_d = torch.empty((P, NUM_NEGATIVES), dtype=torch.int32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    # one wave per pair counts the semi-hard negatives, draws the five positions and maps them to rows (slic_triplet_select_k): no torch
    # cumsum / topk on this loss
    call("slic_triplet_select_k", ptr(Dm), ptr(lab_d), n, ptr(anc_d), ptr(pos_d), P, float(margin), NUM_NEGATIVES, ptr(u), ptr(neg_d),
         ptr(status), stream())
    if int(status.item()) != 0:
        raise RuntimeError("all_semi_hard needs at least {} negatives per anchor (the reference's topk fails the same way)"
                           .format(NUM_NEGATIVES))
    return neg_d.long()                                                  # [P, 5] distinct selectable rows


def get_triplets(embeddings, labels, margin, sampling_strategy, dist_metric='cosine'):
    """NegativeTripletSelector.get_triplets (loss/triplet_loss.py:275-309): [anchor idx], [positive idx], [negative idx]
    as int64 device tensors.  Pairs = combinations of the rows of every label with >= 2 rows, labels in ascending order
    (torch.unique), rows ascending — the reference's enumeration order."""
    assert dist_metric in ('cosine', 'euclidean')
    lab_h = labels.detach().cpu().numpy()
    assert -1 not in lab_h                           # the reference's assert (:285)
    n = len(lab_h)
    anc, pos = _anchor_positive_pairs(lab_h)
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
            if self.dist_metric == 'euclidean':
                return _NTXentEuclid.apply(embeddings, NCE_TEMPERATURE), 0
            return ntxent_loss(embeddings), 0
        if sampling_strategy == 'all_semi_hard':
            if self.dist_metric != 'cosine':
                print('Euclidean dist not supported with infonce loss')
                assert (0)                                   # the reference's own behaviour (:188-190)
            lab_h = labels.detach().cpu().numpy()
            assert -1 not in lab_h
            anc, pos = _anchor_positive_pairs(lab_h)
            if not anc:
                return torch.zeros(1, requires_grad=True), 0                 # :197-198
            negs = all_semi_hard_negatives(embeddings, labels, self.margin, anc, pos)
            dev = embeddings.device
            e = embeddings.float()
            a = torch.as_tensor(anc, dtype=torch.long, device=dev)
            idx = torch.cat([torch.as_tensor(pos, dtype=torch.long, device=dev)[:, None], negs], 1)     # [P, 1 + 5]
            y = e.index_select(0, idx.reshape(-1)).view(len(anc), 1 + NUM_NEGATIVES, -1)
            loss = _InfoNCERows.apply(e.index_select(0, a), y, NCE_TEMPERATURE)
            return loss.reshape(1), len(anc)                                  # the reference returns a 1-element tensor
        if sampling_strategy in ("random_negative", "random_semi_hard", "fixed_semi_hard"):
            a, p, n = get_triplets(embeddings, labels, self.margin, sampling_strategy, self.dist_metric)
            if a.numel() == 0:
                return torch.zeros(1, requires_grad=True).mean(), 0          # loss/triplet_loss.py:222-223
            e = embeddings.float()
            # relu(ap_dists - an_dists + margin).mean() on 1 - cos or F.pairwise_distance (:212-227); index_select's backward
            # scatters the grads
            loss = margin_distance_loss(e.index_select(0, a), e.index_select(0, p), e.index_select(0, n), self.margin, self.dist_metric)
            return loss, int(a.numel())
        raise NotImplementedError(f"sampling_strategy={sampling_strategy!r} ('adapted_hard' belongs to MemTripletLoss)")


def pdist(vectors, eps, dist_metric):
    """cosine (1 - cos) or euclidean distance matrix between all rows (loss/triplet_loss.py:429-437), one kernel"""
    v = vectors.contiguous().float()
    n, D = v.shape
    out = torch.empty(n, n, dtype=torch.float32, device=v.device)
    call("slic_pdist", ptr(v), n, D, float(eps), int(dist_metric == 'euclidean'), ptr(out), stream())
    return out


def pdist_v2(vector1, vector2, eps, dist_metric):
    """rectangular distance matrix [len(vector1), len(vector2)] (loss/triplet_loss.py:439-447), one kernel"""
    x, y = vector1.contiguous().float(), vector2.contiguous().float()
    out = torch.empty(x.shape[0], y.shape[0], dtype=torch.float32, device=x.device)
    call("slic_pdist2", ptr(x), x.shape[0], ptr(y), y.shape[0], x.shape[1], float(eps), int(dist_metric == 'euclidean'), ptr(out),
         stream())
    return out


class MemTripletLoss(nn.Module):
    """loss/triplet_loss.py:9-81: margin-triplet loss whose negatives (and positives' copies) come from a K = 40 slot queue of
    the most recent embeddings.  Same buffers (`queue` [40, 128] of unit rows, `label_q` filled with -1, `queue_ptr`), same
    call: forward(embeddings [b, 128], labels [b], sampling_strategy='adapted_hard') -> (loss, n_triplets).
    Per call: the batch is written into the queue at queue_ptr (all-gathered first when a process group with more than one
    rank exists — the reference tests torch.cuda.device_count() instead and then needs the group anyway), the [b, 40]
    distance matrix is one kernel, every anchor/positive pair of the batch picks a negative among the queue slots of another
    label on the device, and the hinge runs in the fused margin kernel.  Gradients reach the batch rows only (the queue is a
    buffer)."""

    def __init__(self, margin, dist_metric='cosine'):
        super(MemTripletLoss, self).__init__()
        self.K = 40
        self.dim = 128
        self.margin = margin
        self.triplet_selector = None
        self.dist_metric = dist_metric
        self.register_buffer("queue", nn.functional.normalize(torch.randn(self.K, self.dim), dim=1))
        self.register_buffer("label_q", torch.empty(self.K).fill_(-1))
        self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys, labels):
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            from ..misc import distributed_helper as du_helper
            keys, labels = du_helper.all_gather([keys.contiguous(), labels.contiguous()])
        batch_size = keys.shape[0]
        ptr_ = int(self.queue_ptr)
        assert self.K % batch_size == 0, 'self.k needs to be integer multiple of K {}, {}'.format(self.K, batch_size)
        self.queue[ptr_:ptr_ + batch_size, :] = keys
        self.label_q[ptr_:ptr_ + batch_size] = labels.to(self.label_q.dtype)
        self.queue_ptr[0] = (ptr_ + batch_size) % self.K

    def forward(self, embeddings, labels, sampling_strategy="adapted_hard"):
        assert self.dist_metric in ('cosine', 'euclidean')
        if sampling_strategy not in _SELECT_MODES:
            raise NotImplementedError(f"sampling_strategy={sampling_strategy!r}")
        if not embeddings.is_cuda:
            raise _lib.SlicError("MemTripletLoss needs device embeddings (no CPU fallback)")
        dev = embeddings.device
        self._dequeue_and_enqueue(embeddings.detach(), labels)
        batch_size = embeddings.shape[0]
        lab_h = labels.detach().cpu().numpy()
        assert -1 not in lab_h
        # pairs of the BATCH (get_global_triplets, :239-272): a label needs two batch rows and one queue slot of another label
        labq_h = self.label_q.detach().cpu().numpy()
        anc, pos, alab = [], [], []
        ptr_now = int(self.queue_ptr)
        for lab in np.unique(lab_h):
            idx = np.nonzero(lab_h == lab)[0]
            if len(idx) < 2 or not (labq_h != lab).any():
                continue
            for a, p in itertools.combinations(idx.tolist(), 2):
                anc.append(a)
                pos.append((ptr_now - batch_size + p) % self.K)          # the queue slot the positive was just written to (:321-322)
                alab.append(int(lab))
        if not anc:
            return torch.zeros(1, requires_grad=True).mean(), 0
        P = len(anc)
        dist_mat = pdist_v2(embeddings.detach(), self.queue, eps=0, dist_metric=self.dist_metric)
        anc_d = torch.tensor(anc, dtype=torch.int32, device=dev)
        pos_d = torch.tensor(pos, dtype=torch.int32, device=dev)
        alab_d = torch.tensor(alab, dtype=torch.int64, device=dev)
        neg_d = torch.empty(P, dtype=torch.int32, device=dev)
        u = torch.rand(P, dtype=torch.float32, device=dev)
        labq_d = self.label_q.to(torch.int64).contiguous()
        call("slic_triplet_select_cross", ptr(dist_mat), ptr(labq_d), self.K, ptr(anc_d), ptr(alab_d), ptr(pos_d), P,
             float(self.margin), _SELECT_MODES[sampling_strategy], ptr(u), ptr(neg_d), stream())
        self.last_triplets = (anc_d.long(), pos_d.long(), neg_d.long())
        e = embeddings.float()
        q = self.queue.detach()
        loss = margin_distance_loss(e.index_select(0, anc_d.long()), q.index_select(0, pos_d.long()), q.index_select(0, neg_d.long()),
                                    self.margin, self.dist_metric)
        return loss, P
