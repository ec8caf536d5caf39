"""round 5: the B = 8 training step of bench.py — forward + NT-Xent + backward + SGD, ~310 launches — captured ONCE into a
hipGraph and replayed, against the eager step.  Measured: eager 12.31 ms, replay 12.78 ms (the GPU is busy 94.5 % of the eager step: kernel
trace, scripts/r5/prof_b8.sh) — the small batch is bound by its small launches, not by the host; with earlier eager steps of another batch
size in the process (--b32first) torch's capture_end segfaults on this image.  Not shipped; this script is the experiment."""
import faulthandler, os, sys, time
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from video_similarity_search_amd.loss import OnlineTripletLoss


class GraphedStep:
    """fn(): one whole step on static tensors, warmed up on a side stream, captured once, replayed"""

    def __init__(self, fn, warmup=3):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fn()

    def replay(self):
        self.graph.replay()
        return self.out


model, _ = bench.build_model()
model = model.cuda().train()
crit = OnlineTripletLoss(0.2, 'cosine')
opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.5)
if "--b32first" in sys.argv:
    x = torch.randn(32, 3, 16, 112, 112, device="cuda"); lab = torch.arange(16).repeat(2).cuda()
    for _ in range(3):
        l, _ = crit(model(x), lab, sampling_strategy='noise_contrastive'); opt.zero_grad(set_to_none=True); l.backward(); opt.step()
    del x
x8 = torch.randn(8, 3, 16, 112, 112, device="cuda"); lab8 = torch.arange(4).repeat(2).cuda()
def step8():
    l8, _ = crit(model(x8), lab8, sampling_strategy='noise_contrastive'); opt.zero_grad(set_to_none=True); l8.backward(); opt.step()
    return l8.detach()
def timed(fn, w, n):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n
print("eager ms", timed(step8, 3, 10) * 1e3, flush=True)
print("capturing", flush=True)
g = GraphedStep(step8, warmup=2)
print("captured", flush=True)
print("graph ms", timed(g.replay, 3, 20) * 1e3, "loss", float(g.out.item()), flush=True)
