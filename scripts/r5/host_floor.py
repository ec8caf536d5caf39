"""round 5: the HOST's share of a training step — the full-width R3D-18 step on clips so small that the GPU work is negligible (2 x 3 x 16 x 32 x 32:
the same ~310 launches, autograd nodes, optimizer as the B = 8 / B = 32 steps), against the B = 8 and B = 32 steps at 112 x 112.
python scripts/r5/host_floor.py"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench
from video_similarity_search_amd.loss import OnlineTripletLoss
model, _ = bench.build_model(); model = model.cuda().train()
crit = OnlineTripletLoss(0.2, 'cosine'); opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.5)
def run(B, S, n=20):
    x = torch.randn(B, 3, 16, S, S, device="cuda"); lab = torch.arange(B // 2).repeat(2).cuda()
    def step():
        l, _ = crit(model(x), lab, sampling_strategy='noise_contrastive'); opt.zero_grad(set_to_none=True); l.backward(); opt.step()
    for _ in range(4): step()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): step()
    t_issue = (time.time() - t) / n
    torch.cuda.synchronize()
    t_all = (time.time() - t) / n
    return t_issue * 1e3, t_all * 1e3, step
for B, S in ((2, 32), (8, 112), (32, 112)):
    ti, ta, step = run(B, S)
    print(f"B = {B:2d}, {S:3d} x {S:3d}: host issue {ti:6.2f} ms / step, wall {ta:6.2f} ms / step", flush=True)
ti, ta, step = run(2, 32)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3500])
