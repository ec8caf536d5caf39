import sys, os, ctypes
sys.path.insert(0, "/root/repo")
import torch
from video_similarity_search_amd.models.conv_plan import ConvPlan
from video_similarity_search_amd._lib import call, stream
B=32
plan = ConvPlan(128, 128, (3,3,3), (1,1,1), (1,1,1), (8,28,28), "cuda")
x = torch.randn((B,8,28,28,128), device="cuda"); w = torch.randn((128,128,3,3,3), device="cuda")*0.05
wp = plan.pack_fwd(w)
M = B*8*28*28; fl = 2.0*M*128*3456
z = torch.empty((B,8,28,28,128), device="cuda")
for flags, name in [(0,"full"),(32,"all DMA out of range")]:
    a = plan._fwd_args(x, B); a.tap_tab = plan.tap_fwd.data_ptr(); a.wgt = wp.data_ptr(); a.wgt_bytes = wp.numel()*4; a.dst = z.data_ptr(); a.relu = flags
    for v in (17,11,14,18):
        call("slic_conv_gemm", ctypes.byref(a), v, stream()); torch.cuda.synchronize()
        e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): call("slic_conv_gemm", ctypes.byref(a), v, stream())
        e1.record(); torch.cuda.synchronize()
        print(f"{name:28s} v{v}: {fl/(e0.elapsed_time(e1)/5)/1e9:6.1f} TF")
