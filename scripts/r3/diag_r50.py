"""per-parameter gradient error of the tiny Bottleneck model against tests/golden/encoder_options.npz (GPU box)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_oracle_encoder import _options_cases
from video_similarity_search_amd.models import generate_model
from video_similarity_search_amd.loss import OnlineTripletLoss
KW = dict(num_classes=101, n_input_channels=3, conv1_t_size=7, conv1_t_stride=1, projection_head=True, predict_temporal_ds=False,
          spatio_temporal_attention=False, classifier=False, dropout=None)
for tag, shortcut, no_pool, depth, sd, x, g, strided in _options_cases(os.path.join(ROOT, "tests", "golden")):
    if tag != "r50":
        continue
    m = generate_model(depth, widen_factor=0.125, hidden_layer=64, out_dim=32, shortcut_type=shortcut, no_max_pool=no_pool, **KW)
    m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
    m = m.cuda().train()
    emb = m(torch.from_numpy(x).cuda())
    loss, _ = OnlineTripletLoss(0.2, 'cosine')(emb, torch.arange(2).repeat(2).cuda(), sampling_strategy='noise_contrastive')
    loss.backward()
    print("emb err", np.abs(emb.detach().cpu().numpy() - g[f"{tag}/train_emb"]).max())
    for k, p in m.named_parameters():
        ref = g[f"{tag}/grad/{k}"]
        got = strided(p.grad.cpu().numpy())
        print(f"{k:40s} rel {np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30):.2e}  max {np.abs(ref).max():.3e}")
