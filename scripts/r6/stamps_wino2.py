"""round 6: per-workgroup timeline of conv_wino2_kernel at the layer1 shape (B = 32) from in-kernel s_memrealtime stamps (100 MHz) in a
DIAGNOSTIC build (-DSLIC_W2_STAMPS -> csrc/_exp/libslic_w2_stamps.so; the shipped library has no stamps).
    build (here):   bash scripts/r4/ab_wino2.sh build "stamps:-DSLIC_W2_STAMPS"
    run (GPU box):  python scripts/r6/stamps_wino2.py [fwd|dgrad]
Answers: where do the 11-13 % of a launch that the no-epilogue build gives back go — the epilogue's phases, the drain of its stores before the
workgroup may end (one workgroup per CU: the next one cannot start before), or the hand-over between two workgroups of a CU?"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SLIC_LIB_PATH", os.path.join(ROOT, "video_similarity_search_amd", "csrc", "_exp", "libslic_w2_stamps.so"))
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
nwg = 3136 + 8
buf = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
assert lib.slic_debug_set_w2_stamps(buf.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record()
torch.cuda.synchronize()
lib.slic_debug_set_w2_stamps(None)
st = buf.cpu().numpy().reshape(nwg, 16)
st = st[st[:, 1] != 0]
ids = st[:, 0]
xcc, hw = (ids >> 32) & 0xF, ids & 0xFFFFFFFF
cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
t0 = st[:, 1].min()
T = (st[:, 1:11] - t0) / 100.0          # us; columns: 0 start, 1 prologue issued, 2 loop end, 3 nh0 image written, 4 nh0 passes done, 5 nh0 stats done,
                                        #              6 nh1 image written, 7 nh1 passes done, 8 nh1 stats done, 9 stores acknowledged
print(f"{mode}: kernel {e0.elapsed_time(e1) * 1e3:.0f} us by events; {len(st)} workgroups on {len(np.unique(cu))} CUs; span of stamps {T.max():.0f} us")
names = ["prologue (decode + first DMAs issued)", "K loop", "Aw^T + image nh0 + barrier", "passes nh0 (reads, stores issued)", "statistics nh0",
         "image nh1 + barriers", "passes nh1", "statistics nh1", "store drain (vmcnt 0)"]
d = np.diff(T, axis=1)
for i, nm in enumerate(names):
    print(f"  {nm:40s} mean {d[:, i].mean():7.2f} us   p10 {np.percentile(d[:, i], 10):7.2f}   p90 {np.percentile(d[:, i], 90):7.2f}")
print(f"  whole workgroup                          mean {(T[:, 9] - T[:, 0]).mean():7.2f} us;  epilogue (loop end -> stores acknowledged) {(T[:, 9] - T[:, 2]).mean():7.2f} us")
gaps = []
for c in np.unique(cu):
    sel = T[cu == c]
    sel = sel[np.argsort(sel[:, 0])]
    gaps += list(sel[1:, 0] - sel[:-1, 9])
gaps = np.array(gaps)
print(f"  hand-over on a CU (previous workgroup's last stamp -> next workgroup's first): mean {gaps.mean():.2f} us, p10 {np.percentile(gaps, 10):.2f}, "
      f"p50 {np.percentile(gaps, 50):.2f}, p90 {np.percentile(gaps, 90):.2f}  ({len(gaps)} hand-overs)")
per_cu = np.array([np.sum(cu == c) for c in np.unique(cu)])
print(f"  workgroups per CU: min {per_cu.min()} max {per_cu.max()}")
c = np.unique(cu)[len(np.unique(cu)) // 2]
sel = T[cu == c]; sel = sel[np.argsort(sel[:, 0])]
print(f"  CU {c:#x}: start / loop end / last stamp (us) of its workgroups")
for s in sel[:14]:
    print("     " + "  ".join(f"{v:8.2f}" for v in (s[0], s[2], s[9])))
