"""round 5: what the parts of conv_wino2_kernel's epilogue cost at the layer1 shape (B = 32): forward with / without the fused BatchNorm
statistics, data gradient plain / with the fused ReLU mask + BatchNorm-backward sums.  Algorithmic TFLOP/s and ms per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_similarity_search_amd.models.conv_plan import ConvPlan
B, C, N, dims = 32, 64, 64, (16, 56, 56)
k = s1 = (3, 3, 3)
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
p = ConvPlan(C, N, k, (1, 1, 1), (1, 1, 1), dims, "cuda", wino=True, wino2=True, wino2_wgrad=True)
x = torch.randn((B,) + dims + (C,), device="cuda")
w = torch.randn((N, C) + k, device="cuda") * 0.05
wu, wud = p.pack_fwd(w), p.pack_dgrad(w)
dz = torch.randn((B,) + dims + (N,), device="cuda")
mask, zz = torch.randn_like(x), torch.randn_like(x)
mean, invstd = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
fl = 2.0 * B * 16 * 56 * 56 * N * C * 27
for _ in range(40):                       # the chip's clock ramps over the first tens of milliseconds of work: whatever is timed first reads ~10 % slow
    p.forward(x, wu, B, want_stats=True)
torch.cuda.synchronize()
for name, fn in (("fwd, BatchNorm statistics fused", lambda: p.forward(x, wu, B, want_stats=True)),
                 ("fwd, no statistics", lambda: p.forward(x, wu, B, want_stats=False)),
                 ("dgrad, plain", lambda: p.dgrad(dz, wud, B)),
                 ("dgrad, mask + BatchNorm-backward sums fused", lambda: p.dgrad(dz, wud, B, mask=mask, bwd=(zz, mean, invstd)))):
    t = timeit(fn)
    print(f"{name:48s} {t:7.4f} ms  {fl / t / 1e9:6.1f} algorithmic TFLOP/s")
