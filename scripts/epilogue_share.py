"""Gather-GEMM kernel on the layer1 shape with its K loop cut to nk k-tiles: t(nk) = a + b * nk separates the fixed cost per
launch (launch + prologue + epilogue of 12 544 workgroups, ~10 rounds over 1280 residency slots) from the k-loop rate.
Epilogue flavours: plain store; + forward BN partials; data-gradient style (addend + ReLU mask + BN-backward partials)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from video_similarity_search_amd import _lib
from video_similarity_search_amd.models.conv_plan import ConvPlan
from video_similarity_search_amd._lib import call, stream

B = 32
plan = ConvPlan(64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), (16, 56, 56), "cuda", wino=False)
x = torch.randn((B, 16, 56, 56, 64), device="cuda")
w = torch.randn((64, 64, 3, 3, 3), device="cuda") * 0.05
wp = plan.pack_fwd(w)
z = torch.empty((B, 16, 56, 56, 64), device="cuda")
add, msk, bz = torch.randn_like(z), torch.randn_like(z), torch.randn_like(z)
mean, invstd = torch.randn(64, device="cuda"), torch.rand(64, device="cuda") + 0.5
lib = _lib.load()
variants = [int(v) for v in sys.argv[1:]] or [22, 20]
for variant in variants:
    for flavour in ("store", "stats", "dgrad"):
        ts = []
        nks = (2, 8, 16, 32, 54)
        for nk in nks:
            a = plan._fwd_args(x, B)
            a.wgt, a.wgt_bytes, a.dst, a.nchunks = wp.data_ptr(), wp.numel() * 4, z.data_ptr(), nk * 8
            tm = lib.slic_conv_tile_m(ctypes.byref(a), variant)
            part = torch.empty((a.M + tm - 1) // tm, 2, 64, device="cuda")
            if flavour == "stats":
                a.stat_partial = part.data_ptr()
            if flavour == "dgrad":
                a.addend, a.mask_src, a.bwd_z = add.data_ptr(), msk.data_ptr(), bz.data_ptr()
                a.bwd_mean, a.bwd_invstd, a.bwd_partial = mean.data_ptr(), invstd.data_ptr(), part.data_ptr()
            call("slic_conv_gemm", ctypes.byref(a), variant, stream()); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                call("slic_conv_gemm", ctypes.byref(a), variant, stream())
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5)
        b, a0 = np.polyfit(nks, ts, 1)
        fl_tile = 2.0 * a.M * 64 * 32
        print(f"variant {variant} {flavour:6s}: " + " ".join(f"nk={n}:{t:6.3f}" for n, t in zip(nks, ts)) +
              f" ms | fixed {a0:5.3f} ms, {b*1e3:6.2f} us/k-tile = {fl_tile / b / 1e9:6.1f} TFLOP/s in the loop")
