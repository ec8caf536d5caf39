"""
Generates tests/golden/encoder_options.npz in the BUILD container by importing the reference's models/resnet.py with the two
constructor options no shipped config selects: the stem's max-pool (`no_max_pool=False`, models/resnet.py:123, 262-263) and
`shortcut_type='A'` (:213-231).  and the Bottleneck depth 50 (:58-96, 449-450).  Tiny-width models: train-mode forward + noise_contrastive loss + backward, then an eval forward.
    python tests/golden/make_goldens_encoder_options.py
Kept small: the weights and the clips are NOT stored — `draw_case` below (numpy PCG64, oracle.encoder.make_state_dict) is what the
test calls to regenerate them, a checksum of each is stored — and a gradient is stored as at most 4096 evenly strided elements.
"""
import os
import sys
sys.dont_write_bytecode = True
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import encoder as oe                          # noqa: E402
if __name__ == "__main__":
    sys.path.insert(0, "/root/reference")
    from models.resnet import generate_model                 # noqa: E402  (the reference)
    from loss.triplet_loss import OnlineTripletLoss           # noqa: E402
    torch.Tensor.cuda = lambda self, *a, **k: self            # oracle-only shim for hard-coded .cuda()

HERE = os.path.dirname(os.path.abspath(__file__))
WIDEN, HIDDEN, OUT_DIM = 0.125, 64, 32
CASES = (("poolA", "A", False, 18), ("poolB", "B", False, 18), ("r50", "B", True, 50))     # tag, shortcut, no_max_pool, depth
LAYERS = {18: (2, 2, 2, 2), 50: (3, 4, 6, 3)}


def draw_case(rng, shortcut, depth=18):
    """the weights (reference init rules, BN affine perturbed) and the clip batch of one case, in the generator's draw order"""
    sd = oe.make_state_dict(rng, layers=LAYERS[depth], widen=WIDEN, hidden=HIDDEN, out_dim=OUT_DIM, bottleneck=depth >= 50)
    if shortcut == "A":
        sd = {k: v for k, v in sd.items() if ".downsample." not in k}
    for k in sd:
        if k.endswith(("bn1.weight", "bn2.weight", "bn3.weight", "downsample.1.weight")) or k == "bn_proj.weight":
            sd[k] = (1.0 + 0.1 * rng.standard_normal(sd[k].shape)).astype(np.float32)
        if k.endswith(("bn1.bias", "bn2.bias", "bn3.bias", "downsample.1.bias")) or k == "bn_proj.bias":
            sd[k] = (0.1 * rng.standard_normal(sd[k].shape)).astype(np.float32)
    x = rng.standard_normal((4, 3, 8, 48, 48)).astype(np.float32)
    return sd, x


def strided(a, limit=4096):
    """at most ~limit evenly strided elements; the step is kept coprime with 27 (a multiple of 3 would walk one tap of a 3 x 3 x 3 filter)"""
    a = np.asarray(a).reshape(-1)
    step = max(1, -(-a.size // limit))
    while step > 1 and step % 3 == 0:
        step += 1
    return a[::step].copy()


def main():
    rng = np.random.default_rng(57)
    out = {}
    for tag, shortcut, no_pool, depth in CASES:
        sd, x = draw_case(rng, shortcut, depth)
        m = generate_model(depth, hidden_layer=HIDDEN, out_dim=OUT_DIM, num_classes=101, n_input_channels=3, shortcut_type=shortcut,
                           conv1_t_size=7, conv1_t_stride=1, no_max_pool=no_pool, widen_factor=WIDEN, projection_head=True,
                           predict_temporal_ds=False, spatio_temporal_attention=False, classifier=False, dropout=None)
        assert sorted(m.state_dict()) == sorted(sd), set(m.state_dict()) ^ set(sd)
        m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
        m.train()
        emb = m(torch.from_numpy(x))
        loss, _ = OnlineTripletLoss(0.2, 'cosine')(emb, torch.arange(2).repeat(2), sampling_strategy='noise_contrastive')
        loss.backward()
        out[f"{tag}/check"] = np.array([float(np.sum(x, dtype=np.float64)), float(sum(np.sum(v, dtype=np.float64) for v in sd.values()))])
        out[f"{tag}/train_emb"], out[f"{tag}/loss"] = emb.detach().numpy(), loss.detach().numpy()
        for k, p in m.named_parameters():
            out[f"{tag}/grad/{k}"] = strided(p.grad.detach().numpy())
        for k, v in m.state_dict().items():
            if k.endswith(("running_mean", "running_var")):
                out[f"{tag}/after/{k}"] = v.numpy().copy()
        m.eval()
        with torch.no_grad():
            out[f"{tag}/eval_emb"] = m(torch.from_numpy(x)).numpy()
        print(tag, "loss", float(loss), "emb", tuple(emb.shape))
    np.savez_compressed(os.path.join(HERE, "encoder_options.npz"), **out)
    print("encoder_options goldens:", len(out))


if __name__ == "__main__":
    main()
