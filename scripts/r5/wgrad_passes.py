"""round 5 (GPU box): the two-dimensional weight gradient of R3D-18's four stride-1 shapes at B = 32, run alone — for a rocprofv3 kernel
trace of its three kernels (scripts/r5/prof_wgrad_passes.sh) and for the whole op's time by events"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_similarity_search_amd.models.conv_plan import ConvPlan
B = 32
k, s, p = (3, 3, 3), (1, 1, 1), (1, 1, 1)
for name, C, dims in (("l1", 64, (16, 56, 56)), ("l2", 128, (8, 28, 28)), ("l3", 256, (4, 14, 14)), ("l4", 512, (2, 7, 7))):
    plan = ConvPlan(C, C, k, s, p, dims, "cuda", batch=B)
    assert plan.wino2_wgrad
    x = torch.randn((B,) + dims + (C,), device="cuda")
    dz = torch.randn((B,) + dims + (C,), device="cuda")
    dW = torch.empty((C, C) + k, device="cuda")
    for _ in range(3): plan.wgrad(x, dz, B, dW)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): plan.wgrad(x, dz, B, dW)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    fl = 2.0 * B * dims[0] * dims[1] * dims[2] * C * C * 27
    print(f"{name}: {t*1e3:8.1f} us  {fl/t/1e9:6.1f} TF", flush=True)
