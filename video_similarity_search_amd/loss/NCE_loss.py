"""
Drop-in for the memory-bank NCE of the reference's loss/NCE_loss.py:
    NCEAverage(inputSize, outputSize, K, T, momentum)(l, ab, y, idx=None) -> (out_l, out_ab)   <- :10-88
    NCESoftmaxLoss()(x) -> scalar                                                               <- :341-352
    AliasMethod(probs).draw(N)                                                                  <- :246-307
Same buffers (`params`, `memory_l`, `memory_ab`), same cross-wiring (out_ab is scored against memory_l, out_l
against memory_ab), bank rows detached, momentum update + renormalise under no_grad.  The gather+bmm, its
backward, the bank update and the class-0 cross-entropy are HIP kernels (csrc/nce.hip).
The use_softmax=False branch (:54-71: out = exp(score / T / Z) with the constants Z_l / Z_ab fixed on the first call and
kept in `params[2:4]`) and its NCECriterion (:312-337, Eq. 12 of the CMC paper) are not selected by SLIC
(online_train.py:701-710 builds NCEAverage with the default and NCESoftmaxLoss); they are restated on top of the same
score kernels — the exp / log over the [B, K+1] outputs are a few kilobytes of elementwise torch ops.
"""
import math

import torch
from torch import nn

from .. import _lib
from .._lib import call, ptr, stream


class AliasMethod(object):
    """Sampler over `probs`.  The reference builds alias tables on the host and draws with
    index_select + bernoulli (loss/NCE_loss.py:246-307); NCEAverage always passes uniform weights
    (`torch.ones(nLem)`, :14-15), for which every alias-table entry has prob 1 and the draw reduces to
    uniform integers — that is what this class does on the device.  Non-uniform tables are built the
    reference's way and sampled with torch ops (cold path)."""

    def __init__(self, probs):
        probs = probs.clone().float()
        if probs.sum() > 1:
            probs.div_(probs.sum())
        K = len(probs)
        self.n = K
        self.uniform = bool(torch.allclose(probs, torch.full_like(probs, 1.0 / K)))
        self.prob = torch.ones(K)
        self.alias = torch.zeros(K, dtype=torch.long)
        if not self.uniform:
            smaller, larger = [], []
            for kk, p in enumerate(probs):
                self.prob[kk] = K * p
                (smaller if self.prob[kk] < 1.0 else larger).append(kk)
            while len(smaller) > 0 and len(larger) > 0:
                small, large = smaller.pop(), larger.pop()
                self.alias[small] = large
                self.prob[large] = (self.prob[large] - 1.0) + self.prob[small]
                (smaller if self.prob[large] < 1.0 else larger).append(large)
            for last_one in smaller + larger:
                self.prob[last_one] = 1
        self.device = torch.device("cpu")

    def cuda(self):
        self.device = torch.device("cuda")
        self.prob = self.prob.cuda()
        self.alias = self.alias.cuda()

    def draw(self, N):
        kk = torch.randint(0, self.n, (N,), dtype=torch.long, device=self.prob.device)
        if self.uniform:
            return kk
        b = torch.bernoulli(self.prob.index_select(0, kk))
        return kk.mul(b.long()) + self.alias.index_select(0, kk).mul((1 - b).long())


class _BankScores(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, bank, idx, T):
        f = feat.contiguous().float()
        B, D = f.shape
        K1 = idx.shape[1]
        out = torch.empty(B, K1, 1, dtype=torch.float32, device=f.device)
        # NCEAverage updates the bank in place right after scoring, so a training pass keeps the rows as scored
        # (written by the same kernel pass that reads them); under no_grad nothing is kept.
        rows = torch.empty(B * K1, D, dtype=torch.float32, device=f.device) if ctx.needs_input_grad[0] else None
        call("slic_nce_scores_fwd", ptr(bank), ptr(idx), ptr(f), B, K1, D, float(T), ptr(out), ptr(rows), stream())
        ctx.T, ctx.shape, ctx.rows = float(T), (B, K1, D), rows
        return out

    @staticmethod
    def backward(ctx, g):
        B, K1, D = ctx.shape
        g = g.contiguous().float()
        df = torch.empty(B, D, dtype=torch.float32, device=g.device)
        ident = torch.arange(B * K1, dtype=torch.long, device=g.device).view(B, K1)
        call("slic_nce_scores_bwd", ptr(ctx.rows), ptr(ident), ptr(g), B, K1, D, ctx.T, ptr(df), stream())
        ctx.rows = None
        return df, None, None, None


class NCEAverage(nn.Module):
    # outputSize = ndata, inputSize = num of features
    def __init__(self, inputSize, outputSize, K, T=0.07, momentum=0.5, use_softmax=True):
        super(NCEAverage, self).__init__()
        _lib.load()
        self.nLem = outputSize
        self.unigrams = torch.ones(self.nLem)
        self.multinomial = AliasMethod(self.unigrams)
        self.multinomial.cuda()
        self.K = K
        self.use_softmax = use_softmax
        self.register_buffer('params', torch.tensor([K, T, -1, -1, momentum]))
        stdv = 1. / math.sqrt(inputSize / 3)
        self.register_buffer('memory_l', torch.rand(outputSize, inputSize).mul_(2 * stdv).add_(-stdv))
        self.register_buffer('memory_ab', torch.rand(outputSize, inputSize).mul_(2 * stdv).add_(-stdv))

    def forward(self, l, ab, y, idx=None):  # index = y = label
        K, T, momentum = self._host_params()
        batchSize = l.size(0)
        if not l.is_cuda:
            raise _lib.SlicError("NCEAverage needs device tensors (no CPU fallback)")
        if idx is None:
            idx = self.multinomial.draw(batchSize * (self.K + 1)).view(batchSize, -1)
            idx.select(1, 0).copy_(y.data)
        idx = idx.contiguous()
        y = y.contiguous()
        # out_ab scores ab against memory_l, out_l scores l against memory_ab (NCE_loss.py:41-48)
        out_ab = _BankScores.apply(ab, self.memory_l, idx, T)
        out_l = _BankScores.apply(l, self.memory_ab, idx, T)
        if not self.use_softmax:
            # NCE_loss.py:54-71.  The reference sets Z from the RAW dot products (before the division by T): mean * outputSize
            outputSize = self.memory_l.size(0)
            Z_l, Z_ab = self.params[2].item(), self.params[3].item()
            if Z_l < 0:
                self.params[2] = out_l.detach().mean() * T * outputSize
                Z_l = self.params[2].item()
                print("normalization constant Z_l is set to {:.1f}".format(Z_l))
            if Z_ab < 0:
                self.params[3] = out_ab.detach().mean() * T * outputSize
                Z_ab = self.params[3].item()
                print("normalization constant Z_ab is set to {:.1f}".format(Z_ab))
            out_l = torch.exp(torch.div(out_l, Z_l))
            out_ab = torch.exp(torch.div(out_ab, Z_ab))
        with torch.no_grad():   # update memory (NCE_loss.py:73-86)
            B, D = l.shape
            call("slic_nce_bank_update", ptr(self.memory_l), ptr(y), ptr(l.detach().contiguous().float()), B, D,
                 float(momentum), stream())
            call("slic_nce_bank_update", ptr(self.memory_ab), ptr(y), ptr(ab.detach().contiguous().float()), B, D,
                 float(momentum), stream())
        return out_l, out_ab

    def _host_params(self):
        """(K, T, momentum) as Python numbers without a device round trip per step: `params` is a registered buffer (the reference
        reads it with .item() in every forward, NCE_loss.py:27-31); the host copy is refreshed when the buffer is written to"""
        key = (self.params.data_ptr(), self.params._version)
        if getattr(self, "_params_key", None) != key:
            v = self.params.detach().cpu().tolist()
            self._params_host = (int(v[0]), float(v[1]), float(v[4]))
            self._params_key = key
        return self._params_host

    def softmax_loss(self, l, ab, y, idx=None):
        """== NCESoftmaxLoss()(out_l) + NCESoftmaxLoss()(out_ab) for (out_l, out_ab) = self(l, ab, y, idx), banks updated the same
        way — the contrastive step of online_train.py:175-190 as three launches instead of a dozen.  Returns (loss, out_l, out_ab)
        with the outputs detached ([B, K+1, 1], for logging: the reference's loop only uses them through the loss)."""
        if not self.use_softmax:
            raise NotImplementedError("softmax_loss is the use_softmax=True step; NCECriterion goes through forward()")
        if not l.is_cuda:
            raise _lib.SlicError("NCEAverage needs device tensors (no CPU fallback)")
        _, T, momentum = self._host_params()
        B = l.size(0)
        if idx is None:
            idx = self.multinomial.draw(B * (self.K + 1)).view(B, -1)
            idx.select(1, 0).copy_(y.data)
        loss, scores = _FusedContrastStep.apply(l, ab, self.memory_l, self.memory_ab, idx.contiguous(), y.contiguous(), T, momentum)
        return loss, scores[1].unsqueeze(-1), scores[0].unsqueeze(-1)


_NCE_PARTS = 16          # slic_hip.h: SLIC_NCE_PARTS


class _FusedContrastStep(torch.autograd.Function):
    """contrast(l, ab, y) + NCESoftmaxLoss on both outputs as ONE autograd node over three launches (csrc/nce.hip:
    nce_fused_fwd / nce_fused_update / nce_fused_bwd)"""

    @staticmethod
    def forward(ctx, l, ab, mem_l, mem_ab, idx, y, T, momentum):
        l = l.contiguous().float()
        ab = ab.contiguous().float()
        B, D = l.shape
        K1 = idx.shape[1]
        dev = l.device
        scores = torch.empty(2, B, K1, dtype=torch.float32, device=dev)
        rows = torch.empty(2, B, K1, D, dtype=torch.float32, device=dev)
        # [lse | rowloss][side][b], then the SLIC_NCE_PARTS (max, sum-exp) pieces of every score row, then the loss
        nstat = 4 * B
        buf = torch.empty(nstat + 4 * B * _NCE_PARTS + 1, dtype=torch.float32, device=dev)
        stat, part, loss = buf[:nstat].view(2, 2, B), buf[nstat:-1], buf[-1]
        call("slic_nce_fused_fwd", ptr(mem_l), ptr(mem_ab), ptr(l), ptr(ab), ptr(idx), B, K1, D, float(T), ptr(scores), ptr(rows),
             ptr(part), stream())
        call("slic_nce_fused_update", ptr(mem_l), ptr(mem_ab), ptr(y), ptr(l), ptr(ab), B, D, float(momentum), ptr(part), ptr(scores),
             K1, ptr(stat[0]), ptr(stat[1]), ptr(loss), stream())
        ctx.mark_non_differentiable(scores)
        ctx.T, ctx.shape = float(T), (B, K1, D)
        # through save_for_backward: the scores handed back for logging are the buffer the backward reads, and a caller's in-place
        # edit must trip autograd's version check instead of corrupting the gradients; a second backward (retain_graph) works
        ctx.save_for_backward(rows, scores, stat)
        return loss, scores

    @staticmethod
    def backward(ctx, g, _gs):
        B, K1, D = ctx.shape
        rows, scores, stat = ctx.saved_tensors
        df = torch.empty(2, B, D, dtype=torch.float32, device=g.device)
        call("slic_nce_fused_bwd", ptr(rows), ptr(scores), ptr(stat[0]), B, K1, D, ctx.T, ptr(g.contiguous().float()), ptr(df), stream())
        return df[1], df[0], None, None, None, None, None, None


class _SoftmaxCE0(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, K1 = x.shape
        lse = torch.empty(B, dtype=torch.float32, device=x.device)
        rowloss = torch.empty(B, dtype=torch.float32, device=x.device)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        call("slic_softmax_ce0_fwd", ptr(x), B, K1, ptr(lse), ptr(rowloss), ptr(loss), stream())
        ctx.save_for_backward(x, lse)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, lse = ctx.saved_tensors
        B, K1 = x.shape
        dx = torch.empty_like(x)
        call("slic_softmax_ce0_bwd", ptr(x), ptr(lse), B, K1, ptr(g.contiguous().float()), ptr(dx), stream())
        return dx


class NCESoftmaxLoss(nn.Module):
    """Softmax cross-entropy loss (a.k.a., info-NCE loss in CPC paper) against class 0"""

    def __init__(self):
        super(NCESoftmaxLoss, self).__init__()

    def forward(self, x):
        bsz = x.shape[0]
        x = x.reshape(bsz, -1).contiguous().float()      # x.squeeze() of [B, K+1, 1]
        return _SoftmaxCE0.apply(x)


class NCECriterion(nn.Module):
    """Eq. (12) L_NCE of the CMC paper (loss/NCE_loss.py:312-337) on the use_softmax=False outputs [B, K+1, 1]:
    -(sum_b log(P_pos / (P_pos + m Pn + eps)) + sum_{b,k} log(m Pn / (P_neg + m Pn + eps))) / B with Pn = 1 / n_data"""

    def __init__(self, n_data):
        super(NCECriterion, self).__init__()
        self.n_data = n_data

    def forward(self, x):
        eps = 1e-7
        bsz = x.shape[0]
        m = x.size(1) - 1
        Pn = 1 / float(self.n_data)
        P_pos = x.select(1, 0)
        log_D1 = torch.div(P_pos, P_pos.add(m * Pn + eps)).log_()
        P_neg = x.narrow(1, 1, m)
        log_D0 = torch.div(P_neg.clone().fill_(m * Pn), P_neg.add(m * Pn + eps)).log_()
        return - (log_D1.sum(0) + log_D0.view(-1, 1).sum(0)) / bsz
