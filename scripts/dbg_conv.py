"""Times the gather-GEMM variants on the layer2 shape through the raw C ABI (no ConvPlan dispatch).
The ablation numbers quoted in DESIGN.md §6 (operands out of range, LDS + MFMA only ...) were taken with temporary debug
switches compiled into the kernels for that experiment; those switches are not part of the library."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_similarity_search_amd.models.conv_plan import ConvPlan
from video_similarity_search_amd._lib import call, stream

B = 32
plan = ConvPlan(128, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1), (8, 28, 28), "cuda", wino=False)
x = torch.randn((B, 8, 28, 28, 128), device="cuda")
w = torch.randn((128, 128, 3, 3, 3), device="cuda") * 0.05
wp = plan.pack_fwd(w)
M = B * 8 * 28 * 28
fl = 2.0 * M * 128 * 3456
z = torch.empty((B, 8, 28, 28, 128), device="cuda")
a = plan._fwd_args(x, B)
a.wgt, a.wgt_bytes, a.dst = wp.data_ptr(), wp.numel() * 4, z.data_ptr()
for v in (20, 22):
    call("slic_conv_gemm", ctypes.byref(a), v, stream()); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        call("slic_conv_gemm", ctypes.byref(a), v, stream())
    e1.record(); torch.cuda.synchronize()
    print(f"variant {v}: {fl / (e0.elapsed_time(e1) / 5) / 1e9:6.1f} TFLOP/s")
