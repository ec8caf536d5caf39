"""round 5: where the HOST's ~6.5 ms of a training step go (tiny clips: the GPU is idle-fast): forward, loss, backward (autograd thread), optimizer;
and a cProfile of the backward segments run in the autograd thread"""
import os, sys, time, cProfile, pstats, io, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench
from video_similarity_search_amd.loss import OnlineTripletLoss
from video_similarity_search_amd.models import resnet as R
model, _ = bench.build_model(); model = model.cuda().train()
crit = OnlineTripletLoss(0.2, 'cosine'); opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.5)
x = torch.randn(2, 3, 16, 32, 32, device="cuda"); lab = torch.arange(1).repeat(2).cuda()
acc = dict(fwd=0.0, loss=0.0, bwd=0.0, opt=0.0, segb=0.0)
orig = R._SegmentFn.backward
prof = cProfile.Profile()
def timed_backward(ctx, dout):
    t = time.perf_counter()
    if PROFILE: prof.enable()
    r = orig(ctx, dout)
    if PROFILE: prof.disable()
    acc["segb"] += time.perf_counter() - t
    return r
R._SegmentFn.backward = staticmethod(timed_backward)
PROFILE = False
def step():
    t0 = time.perf_counter(); e = model(x)
    t1 = time.perf_counter(); l, _ = crit(e, lab, sampling_strategy='noise_contrastive'); opt.zero_grad(set_to_none=True)
    t2 = time.perf_counter(); l.backward()
    t3 = time.perf_counter(); opt.step()
    t4 = time.perf_counter()
    acc["fwd"] += t1 - t0; acc["loss"] += t2 - t1; acc["bwd"] += t3 - t2; acc["opt"] += t4 - t3
for _ in range(5): step()
torch.cuda.synchronize()
for k in acc: acc[k] = 0.0
n = 30
for _ in range(n): step()
torch.cuda.synchronize()
print({k: round(v / n * 1e3, 3) for k, v in acc.items()}, "ms per step (segb = inside the six segment backwards)")
PROFILE = True
for _ in range(10): step()
torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4200])
