"""round 5 ("what's weak": the plan thresholds were tuned at B = 32 / 8, 112 x 112, width 1): training-step time at shapes
nobody tuned, with the default plan rules against SLIC_WINO2=0 (one-dimensional Winograd + direct) and SLIC_WINO=0 (direct kernels only).
A default that loses to either would be a rule to fix.   python scripts/r5/untuned_shapes_speed.py"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from video_similarity_search_amd.models import generate_model
from video_similarity_search_amd.loss.triplet_loss import ntxent_loss

SHAPES = [(5, 96, 0.5), (13, 128, 1.0), (39, 160, 0.5), (13, 96, 1.0), (5, 160, 1.0), (39, 128, 1.0), (16, 112, 1.0), (24, 128, 1.0), (48, 112, 1.0),
          (64, 112, 1.0), (13, 128, 2.0)]


def step_ms(B, S, widen, env):
    for k in ("SLIC_WINO", "SLIC_WINO2"):
        os.environ.pop(k, None)
    os.environ.update(env)
    kw = dict(bench.R3D18_KW, widen_factor=widen)
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        m = generate_model(18, **kw).cuda().train()
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.5)
    x = torch.randn(B + (B & 1), 3, 16, S, S, device="cuda")

    def step():
        loss = ntxent_loss(m(x))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    eng = m._engine(x)
    kinds = "".join({"wino2": "2", "wino": "1", "direct": "d"}["wino2" if p.wino2 else "wino" if p.wino else "direct"]
                    for (_, p1, p2, _) in eng.blocks for p in (p1, p2))
    del m, opt, x
    torch.cuda.empty_cache()
    return (time.time() - t) / 5 * 1e3, kinds


print(f"{'B':>3s} {'S':>4s} {'width':>5s} | {'default ms':>10s} {'plans (2 = F(4,3)xF(2,3), 1 = F(4,3), d = direct)':50s} | {'WINO2=0':>8s} | {'WINO=0':>8s}")
for B, S, w in SHAPES:
    d, kinds = step_ms(B, S, w, {})
    a, _ = step_ms(B, S, w, {"SLIC_WINO2": "0"})
    b, _ = step_ms(B, S, w, {"SLIC_WINO": "0"})
    flag = "" if d <= 1.02 * min(a, b) else "   <-- default slower"
    print(f"{B:3d} {S:4d} {w:5.2f} | {d:10.2f} {kinds:50s} | {a:8.2f} | {b:8.2f}{flag}", flush=True)
