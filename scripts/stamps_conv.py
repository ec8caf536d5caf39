"""Per-workgroup timelines of the gather-GEMM on the layer1 shape, from in-kernel s_memrealtime stamps (100 MHz) in a
DIAGNOSTIC build of conv.hip (-DSLIC_STAMPS -> csrc/_exp/libslic_stamps.so; the shipped library has no stamps).
    build (here):   python scripts/stamps_conv.py build
    run (GPU box):  python scripts/stamps_conv.py run [variant]
Answers: do the workgroups of one CU run their phases in lockstep (all in the epilogue at once = matrix pipe idle)?"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "video_similarity_search_amd", "csrc")
SO = os.path.join(CS, "_exp", "libslic_stamps.so")

if sys.argv[1] == "build":
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    objs = []
    for f in sorted(os.listdir(CS)):
        if not f.endswith(".hip"):
            continue
        obj = "/tmp/_stamps_" + f[:-4] + ".o"
        extra = ["-ffp-contract=off"] if f == "kmeans.hip" else []
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-DSLIC_STAMPS",
                               "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(CS, f), "-o", obj] + extra)
        objs.append(obj)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs)
    print("built", SO)
    sys.exit(0)

os.environ["SLIC_LIB_PATH"] = SO
sys.path.insert(0, ROOT)
import ctypes
import numpy as np
import torch
from video_similarity_search_amd import _lib
from video_similarity_search_amd.models.conv_plan import ConvPlan
from video_similarity_search_amd._lib import call, stream

variant = int(sys.argv[2]) if len(sys.argv) > 2 else 22
B = 32
plan = ConvPlan(64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), (16, 56, 56), "cuda", wino=False)
x = torch.randn((B, 16, 56, 56, 64), device="cuda")
w = torch.randn((64, 64, 3, 3, 3), device="cuda") * 0.05
wp = plan.pack_fwd(w)
z = torch.empty((B, 16, 56, 56, 64), device="cuda")
lib = _lib.load()
lib.slic_debug_set_stamps.restype = ctypes.c_int
lib.slic_debug_set_stamps.argtypes = [ctypes.c_void_p]
a = plan._fwd_args(x, B)
a.wgt, a.wgt_bytes, a.dst = wp.data_ptr(), wp.numel() * 4, z.data_ptr()
tm = lib.slic_conv_tile_m(ctypes.byref(a), variant)
nwg = ((a.M + tm - 1) // tm + 7) // 8 * 8
for _ in range(3):
    call("slic_conv_gemm", ctypes.byref(a), variant, stream())
torch.cuda.synchronize()
buf = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
assert lib.slic_debug_set_stamps(buf.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
call("slic_conv_gemm", ctypes.byref(a), variant, stream())
e1.record()
torch.cuda.synchronize()
lib.slic_debug_set_stamps(None)
st = buf.cpu().numpy().reshape(nwg, 8)
st = st[st[:, 1] != 0]
ids = st[:, 0]
xcc, hw = (ids >> 32) & 0xF, ids & 0xFFFFFFFF
cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)       # xcc | se | sh | cu
t0 = st[:, 1].min()
T = (st[:, 1:6] - t0) / 100.0                                                                  # microseconds
print(f"variant {variant}: kernel {e0.elapsed_time(e1)*1e3:.0f} us by events; {len(st)} workgroups on {len(np.unique(cu))} CUs; "
      f"span of stamps {T.max():.0f} us")
d = np.diff(T, axis=1)
for name, col in (("prologue issue", 0), ("first tile wait", 1), ("k loop", 2), ("epilogue", 3)):
    print(f"  {name:16s} mean {d[:, col].mean():8.2f} us   p10 {np.percentile(d[:, col], 10):8.2f}   p90 {np.percentile(d[:, col], 90):8.2f}")
print(f"  whole workgroup  mean {(T[:, 4]-T[:, 0]).mean():8.2f} us")
wv, wb, lp = (st[:, 6] >> 32).astype(np.float64), (st[:, 6] & 0xFFFFFFFF).astype(np.float64), st[:, 7].astype(np.float64)
print(f"  wave 0 inside the k loop: {100*(wv/lp).mean():.1f} % of its cycles at the counted vmcnt wait (DMA not landed), "
      f"{100*(wb/lp).mean():.1f} % at the barrier; loop = {lp.mean()/1e3:.1f} k cycles = {lp.mean()/ (T[:,3]-T[:,2]).mean()/1e3:.2f} GHz")
# per CU: fraction of the kernel during which NO resident workgroup is inside its k loop
idle_frac, conc = [], []
for c in np.unique(cu):
    sel = T[cu == c]
    ev = sorted([(s[2], 1) for s in sel] + [(s[3], -1) for s in sel])
    cur, last, idle = 0, 0.0, 0.0
    for t, dlt in ev:
        if cur == 0:
            idle += t - last
        cur += dlt
        last = t
    idle += T.max() - last
    idle_frac.append(idle / T.max())
    conc.append(len(sel))
print(f"  per CU: workgroups {np.mean(conc):.1f} (min {min(conc)}, max {max(conc)}); time with NO workgroup in its k loop: "
      f"mean {100*np.mean(idle_frac):.1f} %  (p10 {100*np.percentile(idle_frac,10):.1f} %, p90 {100*np.percentile(idle_frac,90):.1f} %)")
# timeline of one CU
c = np.unique(cu)[len(np.unique(cu)) // 2]
sel = T[cu == c]
sel = sel[np.argsort(sel[:, 0])]
print(f"  CU {c:#x}: start / loop-begin / loop-end / end (us) of its first 14 workgroups")
for s in sel[:14]:
    print("     " + "  ".join(f"{v:8.1f}" for v in (s[0], s[2], s[3], s[4])))
