"""round 6: per-block timeline of the PERSISTENT kernel conv_wino2p_kernel at the layer1 shape (B = 32), from in-kernel s_memrealtime stamps
(100 MHz) of a diagnostic build (-DSLIC_W2_STAMPS -> csrc/_exp/libslic_w2_stamps.so; the shipped library has none).
    build (here):   bash scripts/r4/ab_wino2.sh build "stamps:-DSLIC_W2_STAMPS"
    run (GPU box):  python scripts/r6/stamps_wino2p.py [fwd|dgrad]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SLIC_LIB_PATH", os.path.join(ROOT, "video_similarity_search_amd", "csrc", "_exp", "libslic_w2_stamps.so"))
os.environ["SLIC_WINO2_PERSIST"] = "1"
sys.path.insert(0, ROOT)
import numpy as np
import torch
from video_similarity_search_amd import _lib
from video_similarity_search_amd.models.conv_plan import ConvPlan

mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
B, C, N, dims = 32, 64, 64, (16, 56, 56)
p = ConvPlan(C, N, (3, 3, 3), (1, 1, 1), (1, 1, 1), dims, "cuda", wino=True, wino2=True, wino2_wgrad=True)
x = torch.randn((B,) + dims + (C,), device="cuda")
w = torch.randn((N, C, 3, 3, 3), device="cuda") * 0.05
wu, wud = p.pack_fwd(w), p.pack_dgrad(w)
dz = torch.randn((B,) + dims + (N,), device="cuda")
mask, zz = torch.randn_like(x), torch.randn_like(x)
mean, invstd = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
fn = (lambda: p.forward(x, wu, B, want_stats=True)) if mode == "fwd" else (lambda: p.dgrad(dz, wud, B, mask=mask, bwd=(zz, mean, invstd)))
lib = _lib.load()
lib.slic_debug_set_w2_stamps.restype = ctypes.c_int
lib.slic_debug_set_w2_stamps.argtypes = [ctypes.c_void_p]
for _ in range(30):
    fn()
torch.cuda.synchronize()
G = 256
buf = torch.zeros(G * 16 * 8, dtype=torch.int64, device="cuda")
assert lib.slic_debug_set_w2_stamps(buf.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record()
torch.cuda.synchronize()
lib.slic_debug_set_w2_stamps(None)
st = buf.cpu().numpy().reshape(G, 16, 8)
ids = st[:, 0, 7]
xcc, hw = (ids >> 32) & 0xF, ids & 0xFFFFFFFF
cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
t0 = st[:, 0, 0][st[:, 0, 0] > 0].min()
print(f"{mode}: kernel {e0.elapsed_time(e1) * 1e3:.0f} us by events; {len(np.unique(cu))} CUs; XCC of blockIdx.x % 8 == its index for "
      f"{np.mean((np.arange(G) % 8) == xcc) * 100:.0f} % of the workgroups")
rows = []
for g in range(G):
    for it in range(16):
        if st[g, it, 5] == 0:
            break
        rows.append([g, it] + [(st[g, it, k] - t0) / 100.0 for k in range(6)])
R = np.array(rows)
T = R[:, 2:]
names = ["K loop (incl. the wait for the first stage)", "decode of the next block, Aw^T, both images, first half's stores, second half's rows read (ring free)",
         "next block's first U stage + pixel double stage issued", "second half's stores (forward) + statistics, first half (lane exchanges)",
         "statistics merged by 64 threads + their stores"]
d = np.diff(T, axis=1)
for i, nm in enumerate(names):
    print(f"  {nm:126s} mean {d[:, i].mean():7.2f} us   p10 {np.percentile(d[:, i], 10):7.2f}   p90 {np.percentile(d[:, i], 90):7.2f}")
per = T[:, 5] - T[:, 0]
print(f"  one block, start to start of the next: mean {per.mean():.2f} us; blocks per workgroup: min {int(R[:, 1].max()) if False else int(min(np.bincount(R[:, 0].astype(int))))} "
      f"max {int(max(np.bincount(R[:, 0].astype(int))))}")
starts = np.array([st[g, 0, 0] for g in range(G)])
print(f"  first stamp of the workgroups: spread {(starts.max() - starts.min()) / 100.0:.2f} us; last stamp: {(T[:, 5].max()):.1f} us after the first")
g = G // 2
sel = R[R[:, 0] == g]
print(f"  workgroup {g}: start / loop end / ring free / end (us) of its blocks")
for s_ in sel:
    print("     " + "  ".join(f"{v:8.2f}" for v in (s_[2], s_[3], s_[4], s_[7])))
