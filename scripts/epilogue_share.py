"""How long does the gather-GEMM kernel take on the layer1 shape when its K loop is cut to ONE k-tile?  (= launch + prologue +
epilogue per workgroup x rounds: an upper bound of what an epilogue overlapped with the next tile's MFMAs could hide)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_similarity_search_amd import _lib
from video_similarity_search_amd.models.conv_plan import ConvPlan
from video_similarity_search_amd._lib import call, stream

B = 32
plan = ConvPlan(64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), (16, 56, 56), "cuda")
x = torch.randn((B, 16, 56, 56, 64), device="cuda")
w = torch.randn((64, 64, 3, 3, 3), device="cuda") * 0.05
wp = plan.pack_fwd(w)
z = torch.empty((B, 16, 56, 56, 64), device="cuda")
lib = _lib.load()
for variant in (22, 20):
    for nch, name in ((plan.nchunks_fwd, "full K = 1728"), (16, "K = 64 (two k-tiles)")):
        a = plan._fwd_args(x, B)
        a.wgt, a.wgt_bytes, a.dst, a.nchunks = wp.data_ptr(), wp.numel() * 4, z.data_ptr(), nch
        tm = lib.slic_conv_tile_m(ctypes.byref(a), variant)
        part = torch.empty((a.M + tm - 1) // tm, 2, 64, device="cuda")
        for stats in (False, True, False, True):
            a.stat_partial = part.data_ptr() if stats else None
            call("slic_conv_gemm", ctypes.byref(a), variant, stream()); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                call("slic_conv_gemm", ctypes.byref(a), variant, stream())
            e1.record(); torch.cuda.synchronize()
            print(f"variant {variant}  {name:22s} stats={int(stats)}: {e0.elapsed_time(e1) / 5:7.3f} ms")
