"""diagnostic (GPU box): where does the B = 32 fc1.weight gradient differ from the fp64 oracle?
separates the backbone's forward error (pooled features) from the head's own arithmetic."""
import contextlib
import io
import sys
import os

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import encoder as oe
from video_similarity_search_amd.models import generate_model
from video_similarity_search_amd.loss.triplet_loss import ntxent_loss

KW = dict(hidden_layer=2048, out_dim=128, num_classes=101, n_input_channels=3, shortcut_type='B',
          conv1_t_size=7, conv1_t_stride=1, no_max_pool=True, widen_factor=1.0, projection_head=True,
          predict_temporal_ds=False, spatio_temporal_attention=False, classifier=False, dropout=None)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rng = np.random.default_rng(7)
sd = oe.make_state_dict(rng)
x = torch.from_numpy(rng.standard_normal((B, 3, 16, 112, 112)).astype(np.float32))
with contextlib.redirect_stdout(io.StringIO()):
    m = generate_model(18, **KW)
    m2 = generate_model(18, **dict(KW, projection_head=False))
m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
m2.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items() if k in m2.state_dict()})
m = m.cuda().train()
emb = m(x.cuda())
loss = ntxent_loss(emb)
loss.backward()
head = ["fc1.weight", "fc1.bias", "bn_proj.weight", "bn_proj.bias", "fc2.weight", "fc2.bias"]
body = ["conv1.weight", "layer1.0.conv1.weight", "layer1.1.conv2.weight", "layer2.0.downsample.0.weight", "layer2.1.conv1.weight",
        "layer3.1.bn2.weight", "layer3.1.conv2.weight", "layer4.0.conv1.weight", "layer4.1.conv1.weight", "layer4.1.conv2.weight",
        "layer4.1.bn2.weight"]
g_gpu = {k: dict(m.named_parameters())[k].grad.cpu() for k in head + body}
emb_gpu = emb.detach().cpu()
del m, emb, loss
torch.cuda.empty_cache()
pooled_gpu = m2.cuda().train()(x.cuda()).detach().cpu()
del m2
torch.cuda.empty_cache()


def head_run(pooled, dtype):
    t = {k: torch.as_tensor(np.asarray(sd[k])).to(dtype).requires_grad_(True) for k in head}
    p = pooled.to(dtype)
    h = F.linear(p, t["fc1.weight"], t["fc1.bias"])
    hb = F.batch_norm(h, None, None, t["bn_proj.weight"], t["bn_proj.bias"], True, 0.1, 1e-5)
    a = F.relu(hb)
    e = F.linear(a, t["fc2.weight"], t["fc2.bias"])
    l = oe.ntxent_loss(e)
    g = torch.autograd.grad(l, [t[k] for k in head])
    return e.detach(), hb.detach(), dict(zip(head, g))


# fp64 pooled from the oracle backbone
t64 = oe.to_torch(sd, dtype=torch.float64)
with torch.no_grad():
    taps = {}
    oe.encoder_forward(t64, x.double(), training=True, taps=taps)
pooled64 = taps["pooled"]
t32 = oe.to_torch(sd)
with torch.no_grad():
    taps32 = {}
    oe.encoder_forward(t32, x, training=True, taps=taps32)
pooled32 = taps32["pooled"]
sc = pooled64.abs().max().item()
print("pooled: |gpu - f64| / max", (pooled_gpu.double() - pooled64).abs().max().item() / sc,
      " |cpu32 - f64| / max", (pooled32.double() - pooled64).abs().max().item() / sc)
e_ref, hb_ref, g_ref = head_run(pooled64, torch.float64)
e_g, hb_g, g_from_gpu_pooled = head_run(pooled_gpu, torch.float64)
e_c, hb_c, g_from_cpu_pooled = head_run(pooled32, torch.float64)
print("emb: gpu vs f64", (emb_gpu.double() - e_ref).abs().max().item(), " f64 head on gpu pooled vs f64", (e_g - e_ref).abs().max().item())
print("relu mask flips in the head: f64 head on gpu pooled", int(((hb_g > 0) != (hb_ref > 0)).sum()),
      " on cpu32 pooled", int(((hb_c > 0) != (hb_ref > 0)).sum()), " min |hb|", hb_ref.abs().min().item())
for k in head:
    s = g_ref[k].abs().max().item()
    print(f"{k:16s} gpu vs f64 {(g_gpu[k].double() - g_ref[k]).abs().max().item() / s:.3e}   "
          f"gpu vs f64-head(gpu pooled) {(g_gpu[k].double() - g_from_gpu_pooled[k]).abs().max().item() / s:.3e}   "
          f"f64-head(gpu pooled) vs f64 {(g_from_gpu_pooled[k] - g_ref[k]).abs().max().item() / s:.3e}   "
          f"f64-head(cpu32 pooled) vs f64 {(g_from_cpu_pooled[k] - g_ref[k]).abs().max().item() / s:.3e}")

# whole-model fp64 / fp32 gradients of the body tensors
t64g = oe.to_torch(sd, dtype=torch.float64, requires_grad=True)
l64 = oe.ntxent_loss(oe.encoder_forward(t64g, x.double(), training=True))
g64 = dict(zip(body, torch.autograd.grad(l64, [t64g[k] for k in body])))
t32g = oe.to_torch(sd, requires_grad=True)
l32 = oe.ntxent_loss(oe.encoder_forward(t32g, x, training=True))
g32 = dict(zip(body, torch.autograd.grad(l32, [t32g[k] for k in body])))
for k in body:
    sc_ = g64[k].abs().max().item()
    print(f"{k:32s} gpu vs f64 {(g_gpu[k].double() - g64[k]).abs().max().item() / sc_:.3e}   cpu32 vs f64 {(g32[k].double() - g64[k]).abs().max().item() / sc_:.3e}")
