"""
The alternate encoder of the reference, R3DNet (VCOP-style R3D; models/r3d/r3d.py:126-187, selected by
cfg.MODEL.ARCH == 'r3d', models/model_utils.py:87-94), on the same HIP plan as the 3D-ResNet.

R3DNet((1, 1, 1, 1)) is a BasicBlock ResNet with one block per stage, a 3x7x7 stem of stride (1, 2, 2), no max-pool and
no projection head: conv -> BN -> ReLU -> conv -> BN -> (+ conv1x1x1/BN shortcut when downsampling) -> ReLU, then a
global average pool to [B, 512].  Only the module / state_dict names differ (conv2..conv5, block1 / blocks.i,
*.temporal_spatial_conv, downsampleconv / downsamplebn), so the classes below are parameter holders with the
reference's names and the forward runs through the engine of models/resnet.py via a small attribute view.
`r3d_model()` restates the wrapper model_utils.py builds around it (backbone + Linear/ReLU/Linear head).
"""
import torch
import torch.nn as nn
from torch.nn.modules.utils import _triple

from .. import _lib
from .._lib import call, ptr, stream
from .conv_plan import ConvPlan
from .resnet import _Engine, _flush_engine_counters, run_engine


class SpatioTemporalConv(nn.Module):
    """models/r3d/r3d.py:10-38: despite the name, one plain nn.Conv3d called `temporal_spatial_conv`"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False):
        super().__init__()
        if bias:
            raise NotImplementedError("R3DNet never builds a biased SpatioTemporalConv")
        self.temporal_spatial_conv = nn.Conv3d(in_channels, out_channels, _triple(kernel_size), stride=_triple(stride),
                                               padding=_triple(padding), bias=False)

    def forward(self, x):
        raise RuntimeError("SpatioTemporalConv is a parameter holder; the encoder runs through R3DNet.forward (HIP plan)")


class SpatioTemporalResBlock(nn.Module):
    """models/r3d/r3d.py:41-90"""

    def __init__(self, in_channels, out_channels, kernel_size, downsample=False):
        super().__init__()
        self.downsample = downsample
        padding = kernel_size // 2
        if downsample:
            self.downsampleconv = SpatioTemporalConv(in_channels, out_channels, 1, stride=2)
            self.downsamplebn = nn.BatchNorm3d(out_channels)
            self.conv1 = SpatioTemporalConv(in_channels, out_channels, kernel_size, padding=padding, stride=2)
        else:
            self.conv1 = SpatioTemporalConv(in_channels, out_channels, kernel_size, padding=padding)
        self.bn1 = nn.BatchNorm3d(out_channels)
        self.relu1 = nn.ReLU()
        self.conv2 = SpatioTemporalConv(out_channels, out_channels, kernel_size, padding=padding)
        self.bn2 = nn.BatchNorm3d(out_channels)
        self.outrelu = nn.ReLU()

    def forward(self, x):
        raise RuntimeError("SpatioTemporalResBlock is a parameter holder; the encoder runs through R3DNet.forward (HIP plan)")


class SpatioTemporalResLayer(nn.Module):
    """models/r3d/r3d.py:93-123"""

    def __init__(self, in_channels, out_channels, kernel_size, layer_size, block_type=SpatioTemporalResBlock, downsample=False):
        super().__init__()
        self.block1 = block_type(in_channels, out_channels, kernel_size, downsample)
        self.blocks = nn.ModuleList([block_type(out_channels, out_channels, kernel_size) for _ in range(layer_size - 1)])

    def forward(self, x):
        raise RuntimeError("SpatioTemporalResLayer is a parameter holder; the encoder runs through R3DNet.forward (HIP plan)")


class _BlockView:
    """a SpatioTemporalResBlock under the attribute names the engine walks (BasicBlock's)"""

    def __init__(self, blk):
        self.conv1, self.bn1 = blk.conv1.temporal_spatial_conv, blk.bn1
        self.conv2, self.bn2 = blk.conv2.temporal_spatial_conv, blk.bn2
        self.downsample = [blk.downsampleconv.temporal_spatial_conv, blk.downsamplebn] if blk.downsample else None
        self._blk = blk


class _LayerView(list):
    def __init__(self, layer):
        super().__init__([_BlockView(layer.block1)] + [_BlockView(b) for b in layer.blocks])
        self._layer = layer

    def parameters(self):
        return self._layer.parameters()


class _NetView:
    def __init__(self, net):
        self.conv1, self.bn1 = net.conv1.temporal_spatial_conv, net.bn1
        self.layer1, self.layer2, self.layer3, self.layer4 = (_LayerView(l) for l in (net.conv2, net.conv3, net.conv4, net.conv5))
        self.no_max_pool = True
        self.projection_head = False


class R3DNet(nn.Module):
    """Same constructor and state_dict as the reference's R3DNet; forward(x[B, 3, T, H, W]) -> [B, 512]."""

    def __init__(self, layer_sizes, block_type=SpatioTemporalResBlock, with_classifier=False, return_conv=False,
                 num_classes=101, modality='rgb'):
        super().__init__()
        if with_classifier or return_conv:
            raise NotImplementedError("SLIC builds R3DNet(with_classifier=False) (models/model_utils.py:90)")
        self.with_classifier, self.return_conv, self.num_classes = with_classifier, return_conv, num_classes
        self.conv1 = SpatioTemporalConv(2 if modality == 'uv' else 3, 64, [3, 7, 7], stride=[1, 2, 2], padding=[1, 3, 3])
        self.bn1 = nn.BatchNorm3d(64)
        self.relu1 = nn.ReLU()
        self.conv2 = SpatioTemporalResLayer(64, 64, 3, layer_sizes[0], block_type=block_type)
        self.conv3 = SpatioTemporalResLayer(64, 128, 3, layer_sizes[1], block_type=block_type, downsample=True)
        self.conv4 = SpatioTemporalResLayer(128, 256, 3, layer_sizes[2], block_type=block_type, downsample=True)
        self.conv5 = SpatioTemporalResLayer(256, 512, 3, layer_sizes[3], block_type=block_type, downsample=True)
        self.pool = nn.AdaptiveAvgPool3d(1)
        self._view = None
        self._engines = {}
        self.register_state_dict_pre_hook(_flush_engine_counters)

    def __getstate__(self):
        # copy.deepcopy(model) / torch.save(model): the module views and execution plans are rebuilt on demand
        state = self.__dict__.copy()
        state["_view"], state["_engines"] = None, {}
        return state

    def forward(self, x):
        if not x.is_cuda:
            raise _lib.SlicError("R3DNet.forward needs a gfx950 device tensor: the encoder has no CPU/PyTorch fallback")
        _lib.load()
        x = x.to(torch.float32)
        if self._view is None or self._view.bn1 is not self.bn1:     # first call, or the modules were swapped (convert_sync_batchnorm)
            self._view = _NetView(self)
            self._engines = {}
        key = (tuple(x.shape), str(x.device))
        eng = self._engines.get(key)
        if eng is None:
            eng = self._engines[key] = _Engine(self._view, x.shape, x.device)
        return run_engine(eng, self, x)


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b on the library's GEMM (slic_conv_gemm: a 1x1x1 convolution on a [B, 1, 1, 1, C] tensor — the route of the ResNet
    head's fc1 / fc2, models/resnet.py), with its data / weight / bias gradients (slic_conv_gemm on the transposed operand, slic_conv_wgrad,
    slic_colsum)."""

    @staticmethod
    def forward(ctx, x, weight, bias, mod):
        B = x.shape[0]
        plan = mod._plan(x.device)
        xc = x.contiguous()
        y, _ = plan.forward(xc.view(B, 1, 1, 1, -1), plan.pack_fwd(weight, fresh=True), B, bias=bias)
        ctx.save_for_backward(xc, weight)
        ctx.mod, ctx.has_bias = mod, bias is not None
        return y.view(B, -1)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        B = x.shape[0]
        plan = ctx.mod._plan(x.device)
        d5 = dy.contiguous().view(B, 1, 1, 1, -1)
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = plan.dgrad(d5, plan.pack_dgrad(weight, fresh=True), B).view(B, -1)
        if ctx.needs_input_grad[1]:
            dW = plan.wgrad(x.view(B, 1, 1, 1, -1), d5, B, torch.empty_like(weight))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(weight.shape[0], dtype=torch.float32, device=x.device)
            call("slic_colsum", ptr(d5), B, weight.shape[0], ptr(db), stream())
        return dx, dW, db, None


class HipLinear(nn.Linear):
    """nn.Linear (same parameters, state_dict keys and initialisation) whose forward / backward run on libslic_hip.so instead of rocBLAS:
    the projection head of r3d_model (models/model_utils.py:90-93).  2-D inputs [B, in_features], in_features % 4 == 0."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__(in_features, out_features, bias=bias)
        assert in_features % 4 == 0, "HipLinear: in_features must be a multiple of 4 (16-byte channel runs)"
        self._plans = {}

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_plans"] = {}
        return state

    def _plan(self, device):
        key = str(device)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = ConvPlan(self.in_features, self.out_features, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), device)
        return plan

    def forward(self, x):
        if not x.is_cuda:
            raise _lib.SlicError("HipLinear.forward needs a gfx950 device tensor: no CPU/PyTorch fallback")
        assert x.dim() == 2 and x.shape[1] == self.in_features, "HipLinear takes [B, in_features]"
        _lib.load()
        return _LinearFn.apply(x.to(torch.float32), self.weight, self.bias, self)


def r3d_model(dim=128, feature_size=512):
    """models/model_utils.py:87-94: nn.Sequential(R3DNet((1,1,1,1)), Linear(512, 512), ReLU, Linear(512, dim)).
    The two head layers run on the library's GEMM like the ResNet head's fc1 / fc2 (HipLinear: an nn.Linear subclass, so the state_dict keys
    1.weight / 1.bias / 3.weight / 3.bias and the optimizer's view of the parameters are the reference's)."""
    return nn.Sequential(R3DNet(layer_sizes=(1, 1, 1, 1), with_classifier=False),
                         HipLinear(feature_size, feature_size), nn.ReLU(), HipLinear(feature_size, dim))
