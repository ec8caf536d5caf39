"""full-size R3D-18 train step at B = 21 (layer2: 1029 workgroups = two rounds + 5: the K-split tail path) with and without the tail split"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from video_similarity_search_amd.models import generate_model
from video_similarity_search_amd.loss.triplet_loss import ntxent_loss
KW = dict(num_classes=101, n_input_channels=3, conv1_t_size=7, conv1_t_stride=1, projection_head=True, predict_temporal_ds=False,
          spatio_temporal_attention=False, classifier=False, dropout=None, shortcut_type='B', no_max_pool=True, hidden_layer=512, out_dim=128)
torch.manual_seed(0)
m = generate_model(18, **KW).cuda().train()
sd = {k: v.clone() for k, v in m.state_dict().items()}
x = torch.randn(22, 3, 16, 112, 112, device="cuda")[:21].contiguous()
res = {}
MODE = os.environ.get("CHECK_MODE", "tail")
for tail in ("1", "0"):
    if MODE == "tail":
        os.environ["SLIC_WINO_TAIL"] = tail
    else:                                   # control: the same network with layer4's K loop cut 8 ways instead of 4 (another summation order)
        os.environ["SLIC_WINO_TAIL"] = "0"
        os.environ["SLIC_WINO_SPLIT"] = "8" if tail == "1" else "1"
    m.load_state_dict(sd)
    m.zero_grad(set_to_none=True)
    x2 = torch.cat([x, x[:1]], 0)          # 22 clips: an even count for NT-Xent
    emb = m(x2)
    ntxent_loss(emb).backward()
    res[tail] = (emb.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()})
e1, g1 = res["1"]; e0, g0 = res["0"]
print("emb max diff", float((e1 - e0).abs().max()), "scale", float(e0.abs().max()))
worst = max(((float((g1[k] - g0[k]).abs().max() / (g0[k].abs().max() + 1e-30)), k) for k in g0
             if float(g0[k].abs().max()) > 1e-5))        # a bias in front of a BatchNorm has a zero gradient: rounding noise
print("worst relative gradient difference", worst)
top = sorted(((float((g1[k] - g0[k]).abs().max() / (g0[k].abs().max() + 1e-30)), k, float(g0[k].abs().max())) for k in g0), reverse=True)[:8]
print(top)
# The gradients of two runs that differ in any summation order sit a few % apart on this random-init network (ReLU inputs within
# rounding of zero take the other branch: CHECK_MODE=control shows the same 4-5 % for a change in layer4's split alone), so only
# the forward is gated here; the tail path itself is unit-tested (tests/test_encoder_gpu.py::test_conv_winograd_f43, 64-64-43).
assert float((e1 - e0).abs().max()) < 1e-4 * max(1.0, float(e0.abs().max()))
print("ok")
