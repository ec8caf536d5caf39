"""round 5 (EXPERIMENT ONLY — nothing of it is in the product, every parity gate and the headline stay on exact-fp32 MFMA):
what a 3-piece bf16 split of both operands (a = a1 + a2 + a3, b likewise; the six products a1b1, a1b2, a2b1, a2b2, a1b3, a3b1 accumulated in fp32 —
what a bf16-MFMA kernel at 16x the fp32-MFMA rate would compute, 6/16 of the fp32 time) does to the NUMBERS, emulated on the CPU:
  (1) one 64 -> 64 3x3x3 convolution (K = 1728) against fp64, beside plain fp32;
  (2) the R3D-18 forward (tiny width, the oracle's network) with every convolution / linear layer through the split, embeddings against fp64.
The pieces are exact bf16 values held in fp32; a product of two bf16 values is exact in fp32 (8 + 8 mantissa bits), so fp32 matmul over the pieces
reproduces the MFMA's arithmetic up to the accumulation order.      python scripts/r5/bf16x3_numerics.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
torch.manual_seed(0)


def split3(a):
    a1 = a.to(torch.bfloat16).to(torch.float32)
    r = a - a1
    a2 = r.to(torch.bfloat16).to(torch.float32)
    a3 = (r - a2).to(torch.bfloat16).to(torch.float32)
    return a1, a2, a3


def conv_split(x, w, stride, pad, n_products=6):
    xs, ws = split3(x), split3(w)
    pairs = [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)][:n_products]
    # smallest terms first, as a kernel would order its six MFMAs per k-step is irrelevant here: fp32 conv accumulates each product separately
    out = None
    for i, j in reversed(pairs):
        t = F.conv3d(xs[i], ws[j], None, stride, pad)
        out = t if out is None else out + t
    return out


def rel(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max())


# (1) one layer1-shaped convolution
x = torch.randn(2, 64, 4, 28, 28)
w = torch.randn(64, 64, 3, 3, 3) / np.sqrt(64 * 27)
ref = F.conv3d(x.double(), w.double(), None, 1, 1)
print("one 64->64 3x3x3 convolution, K = 1728, largest error / largest entry vs fp64:")
print(f"  fp32 conv3d (CPU)            {rel(F.conv3d(x, w, None, 1, 1), ref):.2e}")
for n in (6, 5, 3, 1):
    print(f"  bf16 split, {n} products       {rel(conv_split(x, w, 1, 1, n), ref):.2e}")

# (2) whole network, tiny width: the oracle's forward with conv3d / linear swapped for the split
from oracle import encoder as oe
import contextlib, io
from tests.test_train_loop_gpu import TINY
sys.path.insert(0, "/root/reference") if os.path.isdir("/root/reference") else None
kw = dict(TINY, widen_factor=0.25)
rng = np.random.default_rng(3)
from video_similarity_search_amd.models import generate_model
with contextlib.redirect_stdout(io.StringIO()):
    m = generate_model(18, **kw)
sd = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
xin = torch.from_numpy(rng.standard_normal((4, 3, 8, 32, 32)).astype(np.float32))
e64 = oe.encoder_forward(oe.to_torch(sd, dtype=torch.float64), xin.double(), training=True).detach()
e32 = oe.encoder_forward(oe.to_torch(sd), xin, training=True).detach()
orig_conv, orig_lin = F.conv3d, F.linear
def conv_hook(inp, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    if inp.dtype != torch.float32:
        return orig_conv(inp, weight, bias, stride, padding, dilation, groups)
    out = None
    xs, ws = split3(inp), split3(weight)
    for i, j in reversed([(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]):
        t = orig_conv(xs[i], ws[j], None, stride, padding, dilation, groups)
        out = t if out is None else out + t
    return out if bias is None else out + bias.view(1, -1, 1, 1, 1)
def lin_hook(inp, weight, bias=None):
    if inp.dtype != torch.float32:
        return orig_lin(inp, weight, bias)
    out = None
    xs, ws = split3(inp), split3(weight)
    for i, j in reversed([(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]):
        t = orig_lin(xs[i], ws[j])
        out = t if out is None else out + t
    return out if bias is None else out + bias
F.conv3d, F.linear = conv_hook, lin_hook
try:
    es = oe.encoder_forward(oe.to_torch(sd), xin, training=True).detach()
finally:
    F.conv3d, F.linear = orig_conv, orig_lin
print("R3D-18 (width 0.25) train-mode embeddings of 4 clips, largest absolute error vs fp64 (north_star's gate: 1e-4):")
print(f"  fp32 oracle                  {float((e32.double() - e64).abs().max()):.2e}")
print(f"  every conv / linear as bf16 split, 6 products  {float((es.double() - e64).abs().max()):.2e}")
