"""micro-benchmark (GPU box): every distinct R3D-18 conv shape at batch B: forward / dgrad / wgrad TFLOP/s per tile variant"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_similarity_search_amd.models.conv_plan import ConvPlan
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
only = sys.argv[2] if len(sys.argv) > 2 else None
SHAPES = [  # name, C, N, k, s, p, in dims
    ("c1 stem", 3, 64, (7, 7, 7), (1, 2, 2), (3, 3, 3), (16, 112, 112)),
    ("c2 l1", 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), (16, 56, 56)),
    ("c3 l2.0.c1", 64, 128, (3, 3, 3), (2, 2, 2), (1, 1, 1), (16, 56, 56)),
    ("c4 l2", 128, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1), (8, 28, 28)),
    ("c6 l3.0.c1", 128, 256, (3, 3, 3), (2, 2, 2), (1, 1, 1), (8, 28, 28)),
    ("c7 l3", 256, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1), (4, 14, 14)),
    ("c9 l4.0.c1", 256, 512, (3, 3, 3), (2, 2, 2), (1, 1, 1), (4, 14, 14)),
    ("c10 l4", 512, 512, (3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 7, 7)),
]
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, C, N, k, s, p, dims in SHAPES:
    if only and only not in name: continue
    plan = ConvPlan(C, N, k, s, p, dims, "cuda", wino=False)
    x = torch.randn((B,) + plan.src_dims + (plan.Cs,), device="cuda")
    w = torch.randn((N, C) + k, device="cuda") * 0.05
    wp = plan.pack_fwd(w)
    wd = plan.pack_dgrad(w) if C > 3 else None
    To, Ho, Wo = plan.out_dims
    M = B * To * Ho * Wo
    fl = 2.0 * M * N * C * k[0] * k[1] * k[2]
    dz = torch.randn((B,) + plan.out_dims + (N,), device="cuda")
    dW = torch.empty_like(w)
    line = f"{name:12s} M={M:8d} N={N:4d} K={C*k[0]*k[1]*k[2]:6d} {fl/1e9:7.1f} GF |"
    if os.environ.get("WGONLY"):
        t = timeit(lambda: plan.wgrad(x, dz, B, dW))
        print(f"{name.split()[0]}:{fl/t/1e9:.0f}", end=" ", flush=True)
        continue
    for v in (22, 20):
        t = timeit(lambda: plan.forward(x, wp, B, want_stats=True, variant=v))
        line += f" fwd v{v} {fl/t/1e9:6.1f}"
    if C > 3:
        for v in (22, 20):
            t = timeit(lambda: plan.dgrad(dz, wd, B, variant=v))
            line += f" | dg v{v} {fl/t/1e9:6.1f}"
    t = timeit(lambda: plan.wgrad(x, dz, B, dW))
    line += f" | wg {fl/t/1e9:6.1f} TF"
    if k == (3, 3, 3) and s == (1, 1, 1) and C % 64 == 0 and N % 64 == 0:
        wplan = ConvPlan(C, N, k, s, p, dims, "cuda", wino=True)      # Winograd F(4,3) along W: TFLOP/s of the direct conv's FLOPs
        wu, wud = wplan.pack_fwd(w), wplan.pack_dgrad(w)
        t = timeit(lambda: wplan.forward(x, wu, B, want_stats=True))
        line += f" | wino fwd {fl/t/1e9:6.1f}"
        t = timeit(lambda: wplan.dgrad(dz, wud, B))
        line += f" dg {fl/t/1e9:6.1f}"
        t = timeit(lambda: wplan.wgrad(x, dz, B, dW))
        line += f" wg {fl/t/1e9:6.1f}"
        try:
            w2 = ConvPlan(C, N, k, s, p, dims, "cuda", wino=True, wino2=True, wino2_wgrad=3 * (C // 64) * (N // 64) <= 256)    # F(4,3) x F(2,3) over (W, H)
            wu, wud = w2.pack_fwd(w), w2.pack_dgrad(w)
            t = timeit(lambda: w2.forward(x, wu, B, want_stats=True))
            line += f" | wino2 fwd {fl/t/1e9:6.1f}"
            t = timeit(lambda: w2.dgrad(dz, wud, B))
            line += f" dg {fl/t/1e9:6.1f}"
            if w2.wino2_wgrad:
                t = timeit(lambda: w2.wgrad(x, dz, B, dW))
                line += f" wg {fl/t/1e9:6.1f}"
        except AssertionError:
            pass
    print(line, flush=True)
