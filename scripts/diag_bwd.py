"""diagnostic (GPU box): intermediates of the last block's backward vs fp64 autograd; determinism check"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import encoder as oe
from video_similarity_search_amd.models import generate_model
from video_similarity_search_amd.models import resnet as R
from video_similarity_search_amd.loss.triplet_loss import ntxent_loss
KW = dict(hidden_layer=2048, out_dim=128, num_classes=101, n_input_channels=3, shortcut_type='B', conv1_t_size=7,
          conv1_t_stride=1, no_max_pool=True, widen_factor=1.0, projection_head=True, predict_temporal_ds=False,
          spatio_temporal_attention=False, classifier=False, dropout=None)
rng = np.random.default_rng(7)
sd = oe.make_state_dict(rng)
x = torch.from_numpy(rng.standard_normal((2, 3, 16, 112, 112)).astype(np.float32))
m = generate_model(18, **KW)
m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
m = m.cuda().train()
# fp64 reference with retained grads on taps
t = oe.to_torch(sd, dtype=torch.float64, requires_grad=True)
taps = {}
l = oe.ntxent_loss(oe.encoder_forward(t, x.double(), training=True, taps=taps))
for v in taps.values(): v.retain_grad()
l.backward()
# monkeypatch _bn_bwd and dgrad to record
rec = []
orig = R._Engine._bn_bwd
def spy(dy, out, z, bn, want_g):
    r = orig(dy, out, z, bn, want_g)
    rec.append(dict(dy=dy.clone(), dz=r[0].clone(), dg=r[2].clone(), db=r[3].clone(), C=bn.C))
    return r
R._Engine._bn_bwd = staticmethod(spy)
def run():
    m.zero_grad(set_to_none=True)
    loss = ntxent_loss(m(x.cuda())); loss.backward()
    return {k: p.grad.clone() for k, p in m.named_parameters()}
g1 = run(); rec1 = list(rec); rec.clear()
g2 = run()
nd = [k for k in g1 if not torch.equal(g1[k], g2[k])]
print("non-deterministic grads:", nd[:10], len(nd))
def cmp(name, got, ref):
    got = got.cpu().double()
    if got.dim() == 5: got = got.permute(0, 4, 1, 2, 3)
    print(f"{name:28s} rel err {(got - ref).abs().max().item() / ref.abs().max().item():.3e}  max {ref.abs().max().item():.3e}")
# rec1 order: head bn_proj, then per block (reverse): bn2, bn1, [downsample]
i = 1
for blkname in ["layer4.1", "layer4.0", "layer3.1", "layer3.0"]:
    cmp(blkname + " dOut(in bn2)", rec1[i]["dy"], taps[blkname].grad); 
    cmp(blkname + " da1 (in bn1)", rec1[i + 1]["dy"], taps[blkname + ".a1"].grad)
    cmp(blkname + " dz2", rec1[i]["dz"], taps[blkname + ".z2"].grad)
    i += 3 if blkname.endswith(".0") else 2
