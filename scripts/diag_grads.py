"""diagnostic (GPU box): per-parameter gradient error of the HIP encoder vs fp64 oracle, next to the fp32 CPU oracle's"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import encoder as oe
from video_similarity_search_amd.models import generate_model
from video_similarity_search_amd.loss.triplet_loss import ntxent_loss
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
S = int(sys.argv[2]) if len(sys.argv) > 2 else 112
KW = dict(hidden_layer=2048, out_dim=128, num_classes=101, n_input_channels=3, shortcut_type='B', conv1_t_size=7,
          conv1_t_stride=1, no_max_pool=True, widen_factor=1.0, projection_head=True, predict_temporal_ds=False,
          spatio_temporal_attention=False, classifier=False, dropout=None)
rng = np.random.default_rng(7)
sd = oe.make_state_dict(rng)
x = torch.from_numpy(rng.standard_normal((B, 3, 16, S, S)).astype(np.float32))
m = generate_model(18, **KW)
m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
m = m.cuda().train()
loss = ntxent_loss(m(x.cuda())); loss.backward()
names = [k for k, _ in m.named_parameters()]
res = {}
for dt in (torch.float32, torch.float64):
    t = oe.to_torch(sd, dtype=dt, requires_grad=True)
    l = oe.ntxent_loss(oe.encoder_forward(t, x.to(dt), training=True))
    res[dt] = (l.item(), dict(zip(names, torch.autograd.grad(l, [t[k] for k in names]))))
print("loss gpu", loss.item(), "cpu32", res[torch.float32][0], "cpu64", res[torch.float64][0])
pd = dict(m.named_parameters())
for k in names:
    r = res[torch.float64][1][k]; sc = r.abs().max().item()
    eg = (pd[k].grad.cpu().double() - r).abs().max().item(); ec = (res[torch.float32][1][k].double() - r).abs().max().item()
    flag = "  <<<" if eg > 3 * ec and eg > 1e-4 * sc else ""
    print(f"{k:34s} max|g| {sc:9.3e}  gpu rel {eg/sc:9.2e}  cpu32 rel {ec/sc:9.2e}{flag}")
