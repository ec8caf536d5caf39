import sys, os, ctypes
sys.path.insert(0, "/root/repo")
import torch
from video_similarity_search_amd.models.conv_plan import ConvPlan
from video_similarity_search_amd._lib import call, stream
B=32
plan = ConvPlan(64, 64, (3,3,3), (1,1,1), (1,1,1), (16,56,56), "cuda")
x = torch.randn((B,16,56,56,64), device="cuda"); w = torch.randn((64,64,3,3,3), device="cuda")*0.05
wp = plan.pack_fwd(w)
M = B*16*56*56; fl = 2.0*M*64*1728
z = torch.empty((B,16,56,56,64), device="cuda")
for flags, name in [(0,"full"),(16,"A loads in 64KB window"),(32,"all loads OOB")]:
    a = plan._fwd_args(x, B); a.tap_tab = plan.tap_fwd.data_ptr(); a.wgt = wp.data_ptr(); a.wgt_bytes = wp.numel()*4; a.dst = z.data_ptr(); a.relu = flags
    for v in (11,13):
        call("slic_conv_gemm", ctypes.byref(a), v, stream()); torch.cuda.synchronize()
        e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): call("slic_conv_gemm", ctypes.byref(a), v, stream())
        e1.record(); torch.cuda.synchronize()
        print(f"{name:28s} v{v}: {fl/(e0.elapsed_time(e1)/5)/1e9:6.1f} TF")
