"""
Generates tests/golden/r3d_tiny.npz in the BUILD container by importing the reference's R3DNet
(/root/reference/models/r3d/r3d.py — never at test time; nothing of the reference is copied).

R3DNet((1, 1, 1, 1)) has 14 M parameters (56 MB), too large for a fixture, so the WEIGHTS are not stored: they are
drawn from numpy's PCG64 (`r3d_weights(default_rng(23))`, shared with the tests through tests/golden/r3d_weights.py) in
the reference's own state_dict order; the fixture holds the input clips and the reference's outputs: train-mode
embeddings, NT-Xent loss, every BatchNorm's running statistics after the step, the gradient of every parameter reduced
to (L2 norm, the first 8 entries), and the eval-mode embeddings afterwards.

    python tests/golden/make_goldens_r3d.py
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch
import torch.nn as nn

torch.Tensor.cuda = lambda self, *a, **k: self          # oracle-only shim for the reference's hard-coded .cuda()
nn.Module.cuda = lambda self, *a, **k: self

from models.r3d.r3d import R3DNet                         # noqa: E402  (the reference)
from loss.triplet_loss import OnlineTripletLoss           # noqa: E402
from r3d_weights import r3d_weights                       # noqa: E402


def main():
    m = R3DNet(layer_sizes=(1, 1, 1, 1), with_classifier=False)
    ref_sd = m.state_dict()
    sd = r3d_weights(np.random.default_rng(23))
    assert list(ref_sd) == list(sd), "key order differs from the reference"
    for k in ref_sd:
        assert tuple(ref_sd[k].shape) == tuple(np.asarray(sd[k]).shape), k
    m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
    rng = np.random.default_rng(29)
    x = rng.standard_normal((4, 3, 8, 32, 32)).astype(np.float32)
    out = {"x": x}
    m.train()
    emb = m(torch.from_numpy(x))
    loss, _ = OnlineTripletLoss(0.2, 'cosine')(emb, torch.arange(2).repeat(2), sampling_strategy='noise_contrastive')
    loss.backward()
    out["train/emb"] = emb.detach().numpy()
    out["train/loss"] = loss.detach().numpy()
    for k, p in m.named_parameters():
        g = p.grad.detach().numpy().reshape(-1)
        out["gnorm/" + k] = np.float64(np.linalg.norm(g.astype(np.float64)))
        out["ghead/" + k] = g[:8].copy()
    for k, v in m.state_dict().items():
        if "running" in k or "num_batches" in k:
            out["after/" + k] = v.detach().numpy().copy()
    m.eval()
    with torch.no_grad():
        out["eval/emb"] = m(torch.from_numpy(x)).numpy()
    np.savez_compressed(os.path.join(HERE, "r3d_tiny.npz"), **out)
    print("r3d_tiny: loss", float(loss), "emb", emb.shape, "keys", len(out))


if __name__ == "__main__":
    main()
