"""
Generates tests/golden/encoder_tiny.npz and loss_ntxent.npz in the BUILD container by importing the
reference modules from /root/reference (never at test time; nothing of the reference is copied).
Recipe: SURVEY.md appendix (dont_write_bytecode + .cuda() no-op shim).

    python tests/golden/make_goldens_encoder.py
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch
import torch.nn as nn

torch.Tensor.cuda = lambda self, *a, **k: self          # oracle-only shim for the reference's hard-coded .cuda()
nn.Module.cuda = lambda self, *a, **k: self

from models.resnet import generate_model                 # noqa: E402  (the reference)
from loss.triplet_loss import OnlineTripletLoss           # noqa: E402
from loss.NCE_loss import NCEAverage, NCESoftmaxLoss      # noqa: E402
from oracle import encoder as oe                          # noqa: E402


def tiny_model(sd_np, widen, hidden, out_dim):
    m = generate_model(18, hidden_layer=hidden, out_dim=out_dim, num_classes=101, n_input_channels=3,
                       shortcut_type='B', conv1_t_size=7, conv1_t_stride=1, no_max_pool=True, widen_factor=widen,
                       projection_head=True, predict_temporal_ds=False, spatio_temporal_attention=False,
                       classifier=False, dropout=None)
    ref_sd = m.state_dict()
    assert sorted(ref_sd) == sorted(sd_np), (set(ref_sd) ^ set(sd_np))
    for k in ref_sd:
        assert tuple(ref_sd[k].shape) == tuple(np.asarray(sd_np[k]).shape), k
    m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd_np.items()})
    return m


def main():
    torch.manual_seed(0)
    rng = np.random.default_rng(7)
    widen, hidden, out_dim = 0.125, 64, 32
    sd = oe.make_state_dict(rng, widen=widen, hidden=hidden, out_dim=out_dim)
    # non-trivial BN affine so gamma/beta gradients are exercised
    for k in sd:
        if k.endswith("bn1.weight") or k.endswith("bn2.weight") or k.endswith("downsample.1.weight") or k == "bn_proj.weight":
            sd[k] = (1.0 + 0.1 * rng.standard_normal(sd[k].shape)).astype(np.float32)
        if k.endswith("bn1.bias") or k.endswith("bn2.bias") or k.endswith("downsample.1.bias") or k == "bn_proj.bias":
            sd[k] = (0.1 * rng.standard_normal(sd[k].shape)).astype(np.float32)
    x = rng.standard_normal((4, 3, 8, 32, 32)).astype(np.float32)
    out = {"x": x}
    for k, v in sd.items():
        out["sd/" + k] = v

    # --- the reference: train-mode forward + noise_contrastive loss + backward + one SGD step
    m = tiny_model(sd, widen, hidden, out_dim)
    m.train()
    crit = OnlineTripletLoss(0.2, 'cosine')
    opt = torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.5)
    emb = m(torch.from_numpy(x))
    loss, _ = crit(emb, torch.arange(2).repeat(2), sampling_strategy='noise_contrastive')
    opt.zero_grad()
    loss.backward()
    out["train/emb"] = emb.detach().numpy()
    out["train/loss"] = loss.detach().numpy()
    for k, p in m.named_parameters():
        out["grad/" + k] = p.grad.detach().numpy().copy()
    opt.step()
    for k, v in m.state_dict().items():
        out["after/" + k] = v.detach().numpy().copy()
    # --- eval-mode forward with the updated weights/running stats
    m.eval()
    with torch.no_grad():
        out["eval/emb"] = m(torch.from_numpy(x)).numpy()
    np.savez_compressed(os.path.join(HERE, "encoder_tiny.npz"), **out)
    print("encoder_tiny: loss", float(loss), "emb", emb.shape, "keys", len(out))

    # --- loss goldens
    lo = {}
    for n, D in [(64, 128), (26, 128), (8, 32), (208, 128)]:
        e = rng.standard_normal((n, D)).astype(np.float32) * 2.0
        et = torch.from_numpy(e).requires_grad_(True)
        l, z = crit(et, torch.arange(n // 2).repeat(2), sampling_strategy='noise_contrastive')
        l.backward()
        lo[f"E_{n}_{D}"] = e
        lo[f"loss_{n}_{D}"] = l.detach().numpy()
        lo[f"grad_{n}_{D}"] = et.grad.numpy().copy()
    # tiny-norm rows hit F.cosine_similarity's clamp
    e = rng.standard_normal((8, 16)).astype(np.float32)
    e[3] = 1e-12
    et = torch.from_numpy(e).requires_grad_(True)
    l, _ = crit(et, torch.arange(4).repeat(2), sampling_strategy='noise_contrastive')
    lo["E_tiny"], lo["loss_tiny"] = e, l.detach().numpy()
    # LLC margin term exactly as online_train.py:321-332 writes it (torch library calls, cosine metric, margin .04)
    xa, xn, xf = [torch.from_numpy(rng.standard_normal((13, 128)).astype(np.float32)).requires_grad_(True) for _ in range(3)]
    dist_ap = 1 - torch.nn.functional.cosine_similarity(xa, xn, dim=1)
    dist_an = 1 - torch.nn.functional.cosine_similarity(xa, xf, dim=1)
    llc = torch.nn.MarginRankingLoss(margin=0.04)(dist_ap, dist_an, torch.FloatTensor(dist_ap.size()).fill_(-1))
    llc.backward()
    lo.update(llc_a=xa.detach().numpy(), llc_n=xn.detach().numpy(), llc_f=xf.detach().numpy(), llc_loss=llc.detach().numpy(),
              llc_ga=xa.grad.numpy(), llc_gn=xn.grad.numpy(), llc_gf=xf.grad.numpy())
    # margin-triplet branch with the deterministic 'fixed_semi_hard' selector (loss/triplet_loss.py:205-227, 311-360)
    for tag, n, labs, margin in [("pairs", 32, np.arange(16).repeat(2).reshape(16, 2).T.reshape(-1), 0.2),
                                 ("clusters", 30, rng.integers(0, 5, 30), 0.2),
                                 ("easy", 24, rng.integers(0, 4, 24), 1e-4)]:
        e = rng.standard_normal((n, 128)).astype(np.float32)
        if tag == "easy":                      # well-separated classes -> no semi-hard negatives -> hardest-easy fallback
            cen = rng.standard_normal((4, 128)).astype(np.float32) * 4
            e = cen[labs] + 0.05 * e
        et = torch.from_numpy(e).requires_grad_(True)
        crit_m = OnlineTripletLoss(margin, 'cosine')
        l, nt = crit_m(et, torch.from_numpy(labs.astype(np.int64)), sampling_strategy='fixed_semi_hard')
        l.backward()
        sel = crit_m.triplet_selector.get_triplets(et.detach(), torch.from_numpy(labs.astype(np.int64)))
        lo[f"trip_{tag}_E"], lo[f"trip_{tag}_labels"], lo[f"trip_{tag}_margin"] = e, labs.astype(np.int64), np.float32(margin)
        lo[f"trip_{tag}_loss"], lo[f"trip_{tag}_n"], lo[f"trip_{tag}_grad"] = l.detach().numpy(), np.int64(nt), et.grad.numpy().copy()
        lo[f"trip_{tag}_idx"] = np.array([[int(v) for v in row] for row in sel], np.int64)
    # memory-bank NCE (loss/NCE_loss.py): fixed idx
    B, D, K, ndata = 8, 128, 64, 1000
    nce = NCEAverage(D, ndata, K, 0.07, 0.5)
    ml, mab = nce.memory_l.clone().numpy(), nce.memory_ab.clone().numpy()
    l_f = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)), dim=1).requires_grad_(True)
    ab_f = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)), dim=1).requires_grad_(True)
    y = torch.from_numpy(rng.choice(ndata, B, replace=False).astype(np.int64))
    idx = torch.from_numpy(rng.integers(0, ndata, (B, K + 1)).astype(np.int64))
    idx[:, 0] = y
    out_l, out_ab = nce(l_f, ab_f, y, idx.clone())
    c = NCESoftmaxLoss()
    tot = c(out_l) + c(out_ab)
    tot.backward()
    lo.update(nce_memory_l=ml, nce_memory_ab=mab, nce_l=l_f.detach().numpy(), nce_ab=ab_f.detach().numpy(),
              nce_y=y.numpy(), nce_idx=idx.numpy(), nce_out_l=out_l.detach().numpy(), nce_out_ab=out_ab.detach().numpy(),
              nce_loss=tot.detach().numpy(), nce_grad_l=l_f.grad.numpy(), nce_grad_ab=ab_f.grad.numpy(),
              nce_memory_l_after=nce.memory_l.numpy(), nce_memory_ab_after=nce.memory_ab.numpy())
    np.savez_compressed(os.path.join(HERE, "loss_ntxent.npz"), **lo)
    print("loss goldens:", len(lo))


if __name__ == "__main__":
    main()
