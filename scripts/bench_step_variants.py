"""training-step time with the loss variants of the shipped configs (per GPU, B = 32 / 39 clips of 3x16x112x112):
noise_contrastive (the headline), random_semi_hard (config/default_params.py:58), random_semi_hard + LLC margin term on 39 clips
(BASELINE configs[3]: 13 anchors + 13 positives + 13 second anchors per GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from video_similarity_search_amd.loss import OnlineTripletLoss
from video_similarity_search_amd.loss.triplet_loss import margin_cosine_loss

model, _ = bench.build_model()
model = model.cuda().train()
opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.5)
crit = OnlineTripletLoss(0.2, 'cosine')
rng = np.random.default_rng(7)


def run(name, B, step, n=6):
    for _ in range(2):
        step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n):
        step()
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    print(f"{name:44s} B={B:3d}  {dt*1e3:7.2f} ms/step  {B/dt:7.1f} clips/s", flush=True)


x32 = torch.from_numpy(rng.standard_normal((32, 3, 16, 112, 112)).astype(np.float32)).cuda()
lab32 = torch.arange(16).repeat(2).cuda()
for strat in ("noise_contrastive", "random_semi_hard", "fixed_semi_hard", "random_negative"):
    def step(strat=strat):
        emb = model(x32)
        loss, _ = crit(emb, lab32, sampling_strategy=strat)
        opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    run(strat, 32, step)
x39 = torch.from_numpy(rng.standard_normal((39, 3, 16, 112, 112)).astype(np.float32)).cuda()
lab26 = torch.arange(13).repeat(2).cuda()


def step_llc():
    out = model(x39)
    loss, _ = crit(out[:26], lab26, sampling_strategy="random_semi_hard")
    loss = loss + 1.0 * margin_cosine_loss(out[:13], out[26:39], out[13:26], 0.04)
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()


run("random_semi_hard + LLC (configs[3] per-GPU step)", 39, step_llc)
