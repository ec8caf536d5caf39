"""
Generates tests/golden/loss_variants.npz in the BUILD container by importing the reference's loss/triplet_loss.py
(OnlineTripletLoss 'all_semi_hard' :118-203, MemTripletLoss :9-81) with the oracle-only .cuda() shim.
    python tests/golden/make_goldens_loss_variants.py
Inputs come from numpy's PCG64 so the GPU box regenerates nothing: everything needed is stored.
"""
import os
import random
import sys
sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn as nn
sys.path.insert(0, "/root/reference")
torch.Tensor.cuda = lambda self, *a, **k: self              # oracle-only shim for hard-coded .cuda()
nn.Module.cuda = lambda self, *a, **k: self
from loss.triplet_loss import MemTripletLoss, OnlineTripletLoss, pdist      # noqa: E402  (the reference)

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(41)
out = {}

# ---- all_semi_hard, deterministic regime: well-separated classes, tiny margin -> no pair has more than 5 candidates, so the
# five negatives are the first five rows of the negatives list and Python's random only permutes them
labs = np.repeat(np.arange(4), 6)[rng.permutation(24)]
cen = rng.standard_normal((4, 128)).astype(np.float32) * 4
E = (cen[labs] + 0.05 * rng.standard_normal((24, 128))).astype(np.float32)
et = torch.from_numpy(E).requires_grad_(True)
random.seed(3)
l, n = OnlineTripletLoss(1e-4, 'cosine')(et, torch.from_numpy(labs.astype(np.int64)), sampling_strategy='all_semi_hard')
l.backward()
out.update(ash_det_E=E, ash_det_labels=labs.astype(np.int64), ash_det_margin=np.float32(1e-4), ash_det_loss=l.detach().numpy(),
           ash_det_n=np.int64(n), ash_det_grad=et.grad.numpy().copy())
print("all_semi_hard deterministic:", float(l), n)

# ---- all_semi_hard, crowded regime: random embeddings, margin 0.2 -> dozens of candidates per pair; the reference's draw is
# Python-RNG-bound, so the golden holds what is checkable: the per-pair candidate-list length L and the loss RANGE over seeds
labs2 = rng.integers(0, 6, 40)
E2 = rng.standard_normal((40, 128)).astype(np.float32)
losses = []
for seed in range(8):
    random.seed(seed)
    l2, n2 = OnlineTripletLoss(0.2, 'cosine')(torch.from_numpy(E2), torch.from_numpy(labs2.astype(np.int64)), sampling_strategy='all_semi_hard')
    losses.append(float(l2))
Dm = pdist(torch.from_numpy(E2), eps=0, dist_metric='cosine').numpy()
Ls = []
for lab in np.unique(labs2):
    idx = np.nonzero(labs2 == lab)[0]
    neg = np.nonzero(labs2 != lab)[0]
    for i in range(len(idx)):
        for j in range(i + 1, len(idx)):
            a, p = idx[i], idx[j]
            Ls.append(max(5, int(((Dm[a, p] + 0.2) - Dm[a, neg] > 0).sum())))
out.update(ash_rand_E=E2, ash_rand_labels=labs2.astype(np.int64), ash_rand_margin=np.float32(0.2), ash_rand_losses=np.array(losses),
           ash_rand_n=np.int64(n2), ash_rand_L=np.array(Ls, np.int64))
print("all_semi_hard crowded: pairs", n2, "loss over seeds", min(losses), max(losses), "L", min(Ls), max(Ls))

# ---- MemTripletLoss: three consecutive calls (the queue carries over), the two deterministic strategies
for strat in ("adapted_hard", "fixed_semi_hard"):
    torch.manual_seed(7)
    m = MemTripletLoss(0.2, 'cosine')
    out[f"mem_{strat}_queue0"] = m.queue.clone().numpy()
    for it in range(3):
        e = rng.standard_normal((8, 128)).astype(np.float32)
        labs3 = (np.arange(4).repeat(2)[rng.permutation(8)] + (4 * it if it < 2 else 0)).astype(np.int64)   # call 3 reuses call 1's labels
        et = torch.from_numpy(e).requires_grad_(True)
        l3, n3 = m(et, torch.from_numpy(labs3), sampling_strategy=strat)
        l3.backward()
        out[f"mem_{strat}_E{it}"], out[f"mem_{strat}_labels{it}"] = e, labs3
        out[f"mem_{strat}_loss{it}"], out[f"mem_{strat}_n{it}"] = l3.detach().numpy(), np.int64(n3)
        out[f"mem_{strat}_grad{it}"] = et.grad.numpy().copy()
    out[f"mem_{strat}_queue3"], out[f"mem_{strat}_label_q3"] = m.queue.numpy().copy(), m.label_q.numpy().copy()
    out[f"mem_{strat}_ptr3"] = np.int64(int(m.queue_ptr))
    print("MemTripletLoss", strat, [float(out[f"mem_{strat}_loss{i}"]) for i in range(3)], int(m.queue_ptr))
# ---- memory-bank NCE without softmax (loss/NCE_loss.py:54-71) + NCECriterion (:312-337): two calls (Z fixed by the first)
from loss.NCE_loss import NCEAverage, NCECriterion       # noqa: E402  (the reference)
B, D, K, ndata = 8, 128, 64, 1000
torch.manual_seed(9)
nce = NCEAverage(D, ndata, K, 0.07, 0.5, use_softmax=False)
out["nce0_memory_l"], out["nce0_memory_ab"] = nce.memory_l.clone().numpy(), nce.memory_ab.clone().numpy()
crit = NCECriterion(ndata)
for it in range(2):
    l_f = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)), dim=1).requires_grad_(True)
    ab_f = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)), dim=1).requires_grad_(True)
    y = torch.from_numpy(rng.choice(ndata, B, replace=False).astype(np.int64))
    idx = torch.from_numpy(rng.integers(0, ndata, (B, K + 1)).astype(np.int64))
    idx[:, 0] = y
    o_l, o_ab = nce(l_f, ab_f, y, idx.clone())
    tot = crit(o_l) + crit(o_ab)
    tot.backward()
    out.update({f"nce0_l{it}": l_f.detach().numpy(), f"nce0_ab{it}": ab_f.detach().numpy(), f"nce0_y{it}": y.numpy(), f"nce0_idx{it}": idx.numpy(),
                f"nce0_out_l{it}": o_l.detach().numpy(), f"nce0_out_ab{it}": o_ab.detach().numpy(), f"nce0_loss{it}": tot.detach().numpy(),
                f"nce0_grad_l{it}": l_f.grad.numpy().copy(), f"nce0_grad_ab{it}": ab_f.grad.numpy().copy()})
out["nce0_params"] = nce.params.detach().numpy().copy()
print("NCE without softmax: Z", nce.params[2:4].tolist(), "loss", float(tot))
# ---- LOSS.DIST_METRIC = 'euclidean' (no shipped config selects it): the deterministic strategies of OnlineTripletLoss, three
# MemTripletLoss calls, and the margin terms of triplet_train_epoch (online_train.py:288-360 — torch calls made inline there,
# repeated here verbatim in meaning: F.pairwise_distance(.., 2) / 1 - F.cosine_similarity + MarginRankingLoss(margin)(ap, an, -1))
import torch.nn.functional as F                          # noqa: E402
for n_e, D_e in ((8, 128), (64, 512)):
    Ee = (rng.standard_normal((n_e, D_e)) * 0.3).astype(np.float32)
    et = torch.from_numpy(Ee).requires_grad_(True)
    le, _ = OnlineTripletLoss(0.2, 'euclidean')(et, torch.arange(n_e // 2).repeat(2), sampling_strategy='noise_contrastive')
    le.backward()
    out.update({f"euc_nce_E_{n_e}": Ee, f"euc_nce_loss_{n_e}": le.detach().numpy(), f"euc_nce_grad_{n_e}": et.grad.numpy().copy()})
    print("euclidean noise_contrastive", n_e, float(le))
labs_e = np.repeat(np.arange(5), 4)[rng.permutation(20)].astype(np.int64)
Ee = rng.standard_normal((20, 128)).astype(np.float32)
et = torch.from_numpy(Ee).requires_grad_(True)
le, ne = OnlineTripletLoss(0.5, 'euclidean')(et, torch.from_numpy(labs_e), sampling_strategy='fixed_semi_hard')
le.backward()
out.update(euc_fsh_E=Ee, euc_fsh_labels=labs_e, euc_fsh_margin=np.float32(0.5), euc_fsh_loss=le.detach().numpy(), euc_fsh_n=np.int64(ne),
           euc_fsh_grad=et.grad.numpy().copy())
print("euclidean fixed_semi_hard", float(le), ne)
for strat in ("adapted_hard", "fixed_semi_hard"):
    torch.manual_seed(11)
    m = MemTripletLoss(0.2, 'euclidean')
    out[f"euc_mem_{strat}_queue0"] = m.queue.clone().numpy()
    for it in range(3):
        e = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((8, 128)).astype(np.float32)), dim=1).numpy()
        labs3 = (np.arange(4).repeat(2)[rng.permutation(8)] + (4 * it if it < 2 else 0)).astype(np.int64)
        et = torch.from_numpy(e).requires_grad_(True)
        l3, n3 = m(et, torch.from_numpy(labs3), sampling_strategy=strat)
        l3.backward()
        out[f"euc_mem_{strat}_E{it}"], out[f"euc_mem_{strat}_labels{it}"] = e, labs3
        out[f"euc_mem_{strat}_loss{it}"], out[f"euc_mem_{strat}_n{it}"] = l3.detach().numpy(), np.int64(n3)
        out[f"euc_mem_{strat}_grad{it}"] = et.grad.numpy().copy()
    print("MemTripletLoss euclidean", strat, [float(out[f"euc_mem_{strat}_loss{i}"]) for i in range(3)])
b3 = 6
O = (rng.standard_normal((3 * b3, 128)) * 0.5).astype(np.float32)
for metric in ("cosine", "euclidean"):
    for name, margin, near, far in (("rsp", 0.1, 1, 2), ("llc", 0.3, 2, 1), ("intra", 0.04, 2, 1)):
        ot = torch.from_numpy(O).requires_grad_(True)
        parts = (ot[:b3], ot[b3:2 * b3], ot[2 * b3:])
        if metric == "euclidean":
            dist_ap, dist_an = F.pairwise_distance(parts[0], parts[near], 2), F.pairwise_distance(parts[0], parts[far], 2)
        else:
            dist_ap, dist_an = 1 - F.cosine_similarity(parts[0], parts[near], dim=1), 1 - F.cosine_similarity(parts[0], parts[far], dim=1)
        lt = torch.nn.MarginRankingLoss(margin=margin)(dist_ap, dist_an, torch.FloatTensor(dist_ap.size()).fill_(-1))
        lt.backward()
        out[f"third_{metric}_{name}_loss"], out[f"third_{metric}_{name}_grad"] = lt.detach().numpy(), ot.grad.numpy().copy()
out["third_O"] = O
np.savez_compressed(os.path.join(HERE, "loss_variants.npz"), **out)
print("loss_variants goldens:", len(out))
