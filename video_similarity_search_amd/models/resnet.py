"""
MI355X drop-in for the reference's models/resnet.py (Hara 3D-ResNet, the "R3D-18" of SLIC):

    generate_model(model_depth, **kwargs) -> nn.Module      <- models/resnet.py:436-456
    ResNet / BasicBlock                                      <- models/resnet.py:100-312, 27-57
    ctor kwargs exactly as model_selector passes them        <- models/model_utils.py:37-51

The module tree only HOLDS parameters (torch's own Conv3d / BatchNorm / Linear objects, so the
reference's init rules (:203-210) and its 129 state_dict keys hold by construction, and
checkpoints written by models/model_utils.py:161-176 load unchanged).  forward() never calls those
modules: the whole encoder runs as one hand-written execution plan over the HIP kernels of
libslic_hip.so (csrc/conv.hip, bn.hip), forward AND backward, exposed to autograd as six segment
Functions (stem | layer1-4 | head, so DDP's bucketed all-reduce overlaps the earlier layers' backward)
— NDHWC activations, fp32 MFMA gather-GEMM convs with fused BatchNorm statistics (forward) and fused
ReLU-mask / BatchNorm-backward sums (data gradient), deterministic reductions.  There is no
PyTorch/CPU fallback path.
"""
import os
import weakref

import torch
import torch.nn as nn
from functools import partial

from .. import _lib
from .._lib import call, ptr, stream
from .conv_plan import ConvPlan

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def conv3x3x3(in_planes, out_planes, stride=1):
    return nn.Conv3d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def conv1x1x1(in_planes, out_planes, stride=1):
    return nn.Conv3d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


class BasicBlock(nn.Module):
    """parameter holder with the reference's attribute names (models/resnet.py:27-57)"""
    expansion = 1

    def __init__(self, in_planes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3x3(in_planes, planes, stride)
        self.bn1 = nn.BatchNorm3d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3x3(planes, planes)
        self.bn2 = nn.BatchNorm3d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise RuntimeError("BasicBlock is a parameter holder; the encoder runs through ResNet.forward (HIP plan)")


class Bottleneck(nn.Module):
    """parameter holder with the reference's attribute names (models/resnet.py:58-96): 1x1x1 -> 3x3x3 (stride) -> 1x1x1 (x 4)"""
    expansion = 4

    def __init__(self, in_planes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv1x1x1(in_planes, planes)
        self.bn1 = nn.BatchNorm3d(planes)
        self.conv2 = conv3x3x3(planes, planes, stride)
        self.bn2 = nn.BatchNorm3d(planes)
        self.conv3 = conv1x1x1(planes, planes * self.expansion)
        self.bn3 = nn.BatchNorm3d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise RuntimeError("Bottleneck is a parameter holder; the encoder runs through ResNet.forward (HIP plan)")


# ------------------------------------------------------------------------------------------------
class _Bn:
    """per-call view of one BatchNorm layer: parameters + the statistics this pass produced"""

    def __init__(self, mod):
        self.mod = mod
        self.C = mod.num_features
        self.mean = self.invstd = self.scale = self.shift = None
        # online_train.py:466-468 (cfg.SYNC_BATCH_NORM): torch.nn.SyncBatchNorm.convert_sync_batchnorm swaps the modules; in
        # training the statistics are then taken over all ranks of the module's process group (one all-gather of [2C + 1]
        # doubles forward, one all-reduce of [2C] doubles backward, per layer — SURVEY.md §8e).  Without an initialised
        # process group a SyncBatchNorm module is plain BatchNorm, as in torch.
        self.group = None
        self.sync = False
        if isinstance(mod, nn.SyncBatchNorm) and torch.distributed.is_available() and torch.distributed.is_initialized():
            self.sync = True
            self.group = mod.process_group if mod.process_group is not None else torch.distributed.group.WORLD
        self.n_total = None        # sync: 0-dim double device tensor, the global sample count of this pass
        # eval mode with gradients enabled (the reference's ResNet.forward is an ordinary autograd graph in any mode): the layer
        # normalises with its RUNNING statistics, which are constants of the pass — backward is dz = gamma * invstd * g,
        # dgamma = sum g * xhat, dbeta = sum g
        self.frozen = False


COUNTS = {"bn_bwd": 0, "bn_bwd_fused": 0}     # launches by flavour (diagnostics / tests)


_GRAD_VIEWS = [None]       # the pass's {parameter: DistributedDataParallel bucket view} (misc.distributed_helper.data_parallel), or None


def _grad_out(p):
    """where the gradient of parameter p is written: a FRESH alias of its DistributedDataParallel bucket view when one was handed over
    (autograd installs a gradient nobody else references as .grad without a copy; the reducer then finds .grad aliasing its bucket and
    copies nothing), else a new tensor"""
    gv = _GRAD_VIEWS[0]
    if gv:
        v = gv.get(p)
        if v is not None and v.shape == p.shape and v.device == p.device and v.dtype == p.dtype:
            return v.detach()
    return torch.empty_like(p, memory_format=torch.contiguous_format)


class _Engine:
    """Execution plan of one ResNet at one input shape: conv plans (tables on the device), forward and backward
    passes written out layer by layer.  Mirrors ResNet.forward / BasicBlock.forward of the reference
    (models/resnet.py:255-312, 41-57) and what autograd derives from them."""

    def __init__(self, net, in_shape, device):
        B, C, T, H, W = in_shape
        self.net = net
        self.device = device
        self.in_shape = tuple(in_shape)
        c1 = net.conv1
        self.stem = ConvPlan(C, c1.out_channels, c1.kernel_size, c1.stride, c1.padding, (T, H, W), device)
        dims = self.stem.out_dims
        # nn.MaxPool3d(3, 2, 1) behind the stem (models/resnet.py:262-263) unless no_max_pool (what every shipped config sets)
        self.pool_in = None if net.no_max_pool else tuple(dims)
        if self.pool_in is not None:
            dims = tuple((d - 1) // 2 + 1 for d in dims)
        self.p3 = {}               # Bottleneck block -> plan of its conv3
        self.short_a = {}          # block -> (stride, planes): shortcut_type 'A' (strided positions, zero channels; no parameters)
        self.blocks = []
        self.layer_blocks = []
        for layer in (net.layer1, net.layer2, net.layer3, net.layer4):
            self.layer_blocks.append([])
            for blk in layer:
                assert all(hasattr(blk, a) for a in ("conv1", "bn1", "conv2", "bn2", "downsample"))
                p1 = ConvPlan(blk.conv1.in_channels, blk.conv1.out_channels, blk.conv1.kernel_size, blk.conv1.stride,
                              blk.conv1.padding, dims, device, batch=B)
                p2 = ConvPlan(blk.conv2.in_channels, blk.conv2.out_channels, blk.conv2.kernel_size, blk.conv2.stride,
                              blk.conv2.padding, p1.out_dims, device, batch=B)
                plast = p2
                if hasattr(blk, "conv3"):                              # Bottleneck (depths 50+): a third, 1x1x1 convolution
                    plast = ConvPlan(blk.conv3.in_channels, blk.conv3.out_channels, blk.conv3.kernel_size, blk.conv3.stride,
                                     blk.conv3.padding, p2.out_dims, device)
                    self.p3[blk] = plast
                pd = None
                if isinstance(blk.downsample, partial):
                    kw = blk.downsample.keywords                       # functools.partial(_downsample_basic_block, planes, stride)
                    self.short_a[blk] = (int(kw["stride"]), int(kw["planes"]))
                    assert tuple((d - 1) // kw["stride"] + 1 for d in dims) == tuple(plast.out_dims)
                elif blk.downsample is not None:
                    dc = blk.downsample[0]
                    pd = ConvPlan(dc.in_channels, dc.out_channels, dc.kernel_size, dc.stride, dc.padding, dims, device)
                    assert pd.out_dims == plast.out_dims
                self.blocks.append((blk, p1, p2, pd))
                self.layer_blocks[-1].append((blk, p1, p2, pd))
                dims = plast.out_dims
        self.final_dims = dims
        # Side channel between the autograd nodes of ONE forward pass (keys carry the pass id, so several passes of the same
        # module that are alive at once — Tripletnet's three, gradient accumulation — never see each other's tensors):
        self._pass_id = 0
        self._live = {}        # (pass, si) -> weakref to the saved context of that segment's pending backward
        self._prefused = {}    # (pass, si) -> (data_ptr, shape, partial sums) of a gradient whose ReLU mask + BN sums are done
        self.feat = self.p3.get(self.blocks[-1][0], self.blocks[-1][2]).N
        if net.projection_head:
            self.fc1 = ConvPlan(net.fc1.in_features, net.fc1.out_features, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), device)
            self.fc2 = ConvPlan(net.fc2.in_features, net.fc2.out_features, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1), device)

    # ------------------------------------------------------------------ small helpers
    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def _vec(self, C):
        return torch.empty(C, dtype=torch.float32, device=self.device)

    def prepack(self, with_dgrad):
        """Pack every layer's weights into the GEMM operand layouts on the side stream, at the start of a pass: 42 small
        launches (0.6 ms of a 67 ms training step when they sat in front of each convolution) run beside the clip
        conversion and the stem convolution instead.  The stem packs its own weights inline so that it can start at once;
        everything else waits for the side stream's event the first time a packed operand is needed (_await_packs)."""
        jobs = [(p, m.weight) for blk, p1, p2, pd in self.blocks
                for p, m in ((p1, blk.conv1), (p2, blk.conv2)) + (((pd, blk.downsample[0]),) if pd is not None else ()) +
                (((self.p3[blk], blk.conv3),) if blk in self.p3 else ())]
        if self.net.projection_head:
            jobs += [(self.fc1, self.net.fc1.weight), (self.fc2, self.net.fc2.weight)]
        self.stem.drop_packs()
        for plan, _ in jobs:
            plan.drop_packs()                      # a pack never outlives the pass it was made for
        main, side = torch.cuda.current_stream(), self._side_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)                        # the optimizer's update of these weights, the previous pass's readers
        with torch.cuda.stream(side):
            for plan, w in jobs:
                plan.pack_fwd(w)
                if with_dgrad:
                    plan.pack_dgrad(w)
        self._pack_event = torch.cuda.Event()
        self._pack_event.record(side)

    def _await_packs(self):
        ev = getattr(self, "_pack_event", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._pack_event = None

    def _bn_train(self, bn, part, M):
        """finalize batch statistics (+ running stats, num_batches_tracked) from the conv epilogue's slab"""
        m = bn.mod
        bn.mean, bn.invstd, bn.scale, bn.shift = self._vec(bn.C), self._vec(bn.C), self._vec(bn.C), self._vec(bn.C)
        part, rows = part
        ws = _lib.workspace(_lib.load().slic_bn_finalize_workspace_bytes(part.shape[0], bn.C), part.device, "bn_fin")
        if bn.sync:
            C = bn.C
            W = torch.distributed.get_world_size(bn.group)
            stats = torch.empty(2 * C + 1, dtype=torch.float64, device=part.device)
            call("slic_bn_merge_stats", ptr(part), part.shape[0], rows, C, M, ptr(stats), ptr(ws), stream())
            stats[2 * C] = float(M)
            allst = torch.empty(W, 2 * C + 1, dtype=torch.float64, device=part.device)
            torch.distributed.all_gather_into_tensor(allst, stats, group=bn.group)
            bn.n_total = allst[:, 2 * C].sum()
            call("slic_bn_finalize_sync", ptr(allst), W, C, BN_EPS, BN_MOMENTUM, ptr(m.weight), ptr(m.bias), ptr(bn.mean),
                 ptr(bn.invstd), ptr(bn.scale), ptr(bn.shift), ptr(m.running_mean), ptr(m.running_var), stream())
            self._nbt.append(m.num_batches_tracked)
            return
        call("slic_bn_finalize", ptr(part), part.shape[0], rows, bn.C, M, BN_EPS, BN_MOMENTUM, ptr(m.weight), ptr(m.bias),
             ptr(bn.mean), ptr(bn.invstd), ptr(bn.scale), ptr(bn.shift), ptr(m.running_mean), ptr(m.running_var),
             ptr(ws), stream())
        self._nbt.append(m.num_batches_tracked)        # counters of one segment are bumped by ONE fused add (seg_forward)

    def _bn_eval(self, bn):
        m = bn.mod
        bn.scale, bn.shift = self._vec(bn.C), self._vec(bn.C)
        call("slic_bn_eval_affine", ptr(m.weight), ptr(m.bias), ptr(m.running_mean), ptr(m.running_var), BN_EPS, bn.C,
             ptr(bn.scale), ptr(bn.shift), stream())

    def _bn_frozen(self, bn):
        """running statistics as the pass's (mean, invstd) + the eval affine; nothing is updated"""
        m = bn.mod
        self._bn_eval(bn)
        bn.mean = m.running_mean.detach().float().contiguous()
        bn.invstd = torch.rsqrt(m.running_var.detach().float() + BN_EPS)
        bn.frozen = True

    @staticmethod
    def _apply(z, bn, res, relu):
        y = torch.empty_like(z)
        M = z.numel() // bn.C
        call("slic_bn_apply", ptr(z), ptr(bn.scale), ptr(bn.shift), ptr(res), int(relu), M, bn.C, ptr(y), stream())
        return y

    @staticmethod
    def _bn_bwd(dy, out, z, bn, want_g):
        """returns (dz, g or None, dgamma, dbeta)"""
        COUNTS["bn_bwd"] += 1
        lib = _lib.load()
        M = z.numel() // bn.C
        dz = torch.empty_like(z)
        g = torch.empty_like(z) if want_g else None
        dgamma, dbeta = _grad_out(bn.mod.weight), _grad_out(bn.mod.bias)
        if bn.sync or bn.frozen:
            if out is None:      # no ReLU to undo: the gradient that reaches z's BatchNorm IS dy (slic_bn_bwd_sums writes no g then)
                _Engine._bn_bwd_sync(None, dy, None, z, bn, None, dz, dgamma, dbeta)
                return dz, (dy if want_g else None), dgamma, dbeta
            gm = g if g is not None else torch.empty_like(z)                      # the masked gradient must exist for phase 2
            _Engine._bn_bwd_sync(None, dy, out, z, bn, gm, dz, dgamma, dbeta)
            return dz, g, dgamma, dbeta
        need_buf = int(out is not None and not want_g)
        ws = _lib.workspace(lib.slic_bn_bwd_workspace_bytes(M, bn.C, need_buf), z.device, "bn_bwd")
        call("slic_bn_bwd", ptr(dy), ptr(out), ptr(z), ptr(bn.mean), ptr(bn.invstd), ptr(bn.mod.weight), M, bn.C,
             ptr(g), ptr(dz), ptr(dgamma), ptr(dbeta), ptr(ws), stream())
        return dz, g, dgamma, dbeta

    @staticmethod
    def _bn_bwd_sync(part, dy, out, z, bn, g, dz, dgamma, dbeta):
        """SyncBatchNorm backward: rank-local sums (from a dgrad epilogue's slab `part`, or from dy / out / z with the masked
        gradient written to g) -> all-reduce over the group -> / global count -> dz.  dgamma, dbeta stay rank-local
        (DistributedDataParallel averages parameter gradients, as with torch's SyncBatchNorm)."""
        lib = _lib.load()
        C = bn.C
        M = z.numel() // C
        R = part.shape[0] if part is not None else 0
        sums = torch.empty(2 * C, dtype=torch.float64, device=z.device)
        ws = _lib.workspace(lib.slic_bn_bwd_sums_workspace_bytes(M, C, R), z.device, "bn_bwd_sums")
        gin = g if part is not None else (g if out is not None else dy)      # what phase 2 reads as the gradient
        call("slic_bn_bwd_sums", ptr(part), R, ptr(dy), ptr(out), ptr(z), ptr(bn.mean), ptr(bn.invstd), M, C,
             ptr(g) if (part is None and out is not None) else None, ptr(sums), ptr(dgamma), ptr(dbeta), ptr(ws), stream())
        if bn.frozen:            # frozen statistics carry no gradient: no mean terms (and nothing to exchange)
            k = torch.zeros(2 * C, dtype=torch.float64, device=z.device)
        else:
            torch.distributed.all_reduce(sums, group=bn.group)
            k = sums / bn.n_total
        call("slic_bn_bwd_apply", ptr(gin), ptr(z), ptr(bn.mean), ptr(bn.invstd), ptr(bn.mod.weight), ptr(k[:C]), ptr(k[C:]),
             M, C, ptr(dz), stream())

    @staticmethod
    def _bn_bwd_fused(part, g, z, bn):
        """second half of _bn_bwd when the dgrad that produced g already masked it and emitted the partial sums;
        returns (dz, dgamma, dbeta)"""
        COUNTS["bn_bwd_fused"] += 1
        lib = _lib.load()
        M = z.numel() // bn.C
        R = part.shape[0]
        dz = torch.empty_like(z)
        dgamma, dbeta = _grad_out(bn.mod.weight), _grad_out(bn.mod.bias)
        if bn.sync or bn.frozen:
            _Engine._bn_bwd_sync(part, None, None, z, bn, g, dz, dgamma, dbeta)
            return dz, dgamma, dbeta
        ws = _lib.workspace(lib.slic_bn_bwd_fused_workspace_bytes(R, bn.C), z.device, "bn_bwd_fused")
        call("slic_bn_bwd_fused", ptr(part), R, ptr(g), ptr(z), ptr(bn.mean), ptr(bn.invstd), ptr(bn.mod.weight), M, bn.C,
             ptr(dz), ptr(dgamma), ptr(dbeta), ptr(ws), stream())
        return dz, dgamma, dbeta

    # ------------------------------------------------------------------ segments
    # The plan is cut into segments — stem | layer1 | layer2 | layer3 | layer4 | head — each exposed to autograd as its
    # own node (_SegmentFn).  Backward then hands the last layers' gradients to autograd first, so under
    # DistributedDataParallel the bucketed RCCL all-reduce of those gradients overlaps with the backward of the
    # earlier, heavier layers (the reference gets the same overlap from DDP over per-module autograd nodes).
    N_SEG = 6

    def seg_params(self, si):
        net = self.net
        if si == 0:
            mods = [net.conv1, net.bn1]
        elif si <= 4:
            mods = [getattr(net, f"layer{si}")]
        else:
            mods = [net.fc1, net.bn_proj, net.fc2] if net.projection_head else []
        return [p for m in mods for p in m.parameters() if p.requires_grad]

    def _conv_bn_act(self, plan, inp, weight, bnmod, res, relu, training, B, keep=False):
        """conv -> BN -> (+res) -> relu; returns (z or None, y, bn).  keep: an eval-mode pass whose backward will run — the
        BatchNorm is applied unfolded on its running statistics, z is kept"""
        bn = _Bn(bnmod)
        wp = plan.pack_fwd(weight)
        if not training and keep:
            z, _ = plan.forward(inp, wp, B)
            self._bn_frozen(bn)
            return z, self._apply(z, bn, res, relu), bn
        if training:
            z, part = plan.forward(inp, wp, B, want_stats=True)
            self._bn_train(bn, part, z.numel() // bn.C)
            y = self._apply(z, bn, res, relu)
            return z, y, bn
        self._bn_eval(bn)                     # eval: BN folded into the conv epilogue
        y, _ = plan.forward(inp, wp, B, scale=bn.scale, shift=bn.shift, addend=res, relu=relu)
        return None, y, bn

    def seg_forward(self, si, inp, training, save):
        """returns (out, ctx).  Segment 0 takes the NCDHW clip batch; 1..4 take/return NDHWC activations;
        5 returns the [B, out_dim] (or [B, 512]) embedding."""
        pend = self.__dict__.setdefault("_nbt_pending", [])
        if si == 0 and pend:                           # a pass that never reached its head segment: its counters are bumped now
            torch._foreach_add_(pend, 1)
            pend.clear()
        self._nbt = []
        try:
            out = self._seg_forward(si, inp, training, save)
        except BaseException:
            # a forward that aborts after some BatchNorm layers have updated their running statistics: their counters are bumped NOW, so that a
            # checkpoint written from the handler agrees with its own running statistics (the reference bumps inside each BatchNorm forward)
            pend.extend(self._nbt)
            self._nbt = []
            self.flush_counters()
            raise
        # num_batches_tracked += 1 for every BatchNorm of the pass by ONE fused add, at the head segment (a tiny launch between two
        # dependent ones costs the forward's chain ~9 us, scripts/r5/ab_bn_merges.sh; per segment that was six of them)
        pend.extend(self._nbt)
        self._nbt = []
        if pend and si == self.N_SEG - 1:
            torch._foreach_add_(pend, 1)
            pend.clear()
        return out

    def flush_counters(self):
        """bump the num_batches_tracked counters a pass has not bumped yet (called when a pass aborts, and by the module's state_dict hook: saved
        counters always match the saved running statistics)"""
        pend = self.__dict__.get("_nbt_pending")
        if pend:
            torch._foreach_add_(pend, 1)
            pend.clear()

    def _seg_forward(self, si, inp, training, save):
        net = self.net
        B = inp.shape[0]
        dev = inp.device
        if si == 0:
            x4 = self.stem.make_source(inp)          # NCDHW clip -> the stem plan's operand layout (W-run for RGB)
            z0, a, bn0 = self._conv_bn_act(self.stem, x4, net.conv1.weight, net.bn1, None, True, training, B, keep=save)
            if self.pool_in is None:
                return a, (dict(x4=x4, z0=z0, a0=a, bn0=bn0) if save else None)
            T, H, W = self.pool_in
            C = a.shape[-1]
            pooled = torch.empty((B,) + tuple((d - 1) // 2 + 1 for d in self.pool_in) + (C,), dtype=torch.float32, device=dev)
            arg = torch.empty(pooled.shape, dtype=torch.int32, device=dev) if save else None
            call("slic_maxpool3d_fwd", ptr(a), B, T, H, W, C, ptr(pooled), ptr(arg), stream())
            return pooled, (dict(x4=x4, z0=z0, a0=a, bn0=bn0, pool_arg=arg) if save else None)
        self._await_packs()
        if si <= 4:
            a = inp
            saved = []
            for blk, p1, p2, pd in self.layer_blocks[si - 1]:
                xin = a
                p3 = self.p3.get(blk)
                z1, a1, b1 = self._conv_bn_act(p1, xin, blk.conv1.weight, blk.bn1, None, True, training, B, keep=save)
                if p3 is not None:
                    z2, a2, b2 = self._conv_bn_act(p2, a1, blk.conv2.weight, blk.bn2, None, True, training, B, keep=save)
                if pd is not None:
                    zd, r, bd = self._conv_bn_act(pd, xin, blk.downsample[0].weight, blk.downsample[1], None, False, training, B, keep=save)
                elif blk in self.short_a:
                    stride, planes = self.short_a[blk]
                    T, H, W = p1.in_dims
                    r = torch.empty((B,) + tuple((p3 or p2).out_dims) + (planes,), dtype=torch.float32, device=dev)
                    call("slic_shortcut_a", ptr(xin), B, T, H, W, xin.shape[-1], stride, planes, ptr(r), stream())
                    zd, bd = None, None
                else:
                    zd, r, bd = None, xin, None
                if p3 is not None:
                    z3, out, b3 = self._conv_bn_act(p3, a2, blk.conv3.weight, blk.bn3, r, True, training, B, keep=save)
                    if save:     # zl / bl: the block's LAST convolution output and BatchNorm (what the layer above fuses into its dgrad)
                        saved.append(dict(x=xin, z1=z1, a1=a1, b1=b1, z2=z2, a2=a2, b2=b2, z3=z3, b3=b3, out=out, zd=zd, bd=bd, zl=z3, bl=b3))
                else:
                    z2, out, b2 = self._conv_bn_act(p2, a1, blk.conv2.weight, blk.bn2, r, True, training, B, keep=save)
                    if save:
                        saved.append(dict(x=xin, z1=z1, a1=a1, b1=b1, z2=z2, out=out, b2=b2, zd=zd, bd=bd, zl=z2, bl=b2))
                a = out
            return a, (dict(blocks=saved) if save else None)
        # head: pool -> fc2(relu(bn_proj(fc1(x))))   (models/resnet.py:286-299)
        a = inp
        To, Ho, Wo = self.final_dims
        S = To * Ho * Wo
        pooled = torch.empty(B, self.feat, dtype=torch.float32, device=dev)
        call("slic_avgpool_fwd", ptr(a), B, S, self.feat, ptr(pooled), stream())
        ctx = dict(last_shape=tuple(a.shape), pooled=pooled) if save else None
        if not net.projection_head:
            return pooled, ctx
        bnp = _Bn(net.bn_proj)
        w1 = self.fc1.pack_fwd(net.fc1.weight)
        p5 = pooled.view(B, 1, 1, 1, self.feat)
        if training:
            h1, part = self.fc1.forward(p5, w1, B, bias=net.fc1.bias, want_stats=True)
            self._bn_train(bnp, part, B)
            ah = self._apply(h1, bnp, None, True)
        elif save:             # eval mode with a backward to come: unfolded, on the running statistics
            h1, _ = self.fc1.forward(p5, w1, B, bias=net.fc1.bias)
            self._bn_frozen(bnp)
            ah = self._apply(h1, bnp, None, True)
        else:
            self._bn_eval(bnp)
            h1 = None      # bias first, then the BN affine, then ReLU — all in the epilogue
            ah, _ = self.fc1.forward(p5, w1, B, bias=net.fc1.bias, scale=bnp.scale, shift=bnp.shift, relu=True)
        w2 = self.fc2.pack_fwd(net.fc2.weight)
        y, _ = self.fc2.forward(ah.view(B, 1, 1, 1, -1), w2, B, bias=net.fc2.bias)
        if save:
            ctx.update(h1=h1, ah=ah, bnp=bnp)
        return y.view(B, -1), ctx

    def _live_ctx(self, pid, si):
        ref = self._live.get((pid, si))
        holder = ref() if ref is not None else None
        return holder.d if holder is not None else None

    def seg_backward(self, si, ctx, dout, pid=None):
        """returns (gradient wrt the segment input or None, {parameter: gradient}); pid = forward pass the context is from"""
        net = self.net
        grads = {}
        B = dout.shape[0]
        dout = dout.contiguous()
        self._await_packs()

        _GRAD_VIEWS[0] = getattr(self, "grad_views", None)
        new_like = _grad_out

        def bias_grad(d2, p):
            g = new_like(p)
            call("slic_colsum", ptr(d2), d2.shape[0], d2.shape[1], ptr(g), stream())
            return g

        # Weight gradients on a side stream: they depend only on (saved activation, dz), nothing downstream in this segment
        # depends on them, and they are MFMA-bound — so they overlap the HBM-bound BatchNorm passes and fill the partial
        # last rounds of the data-gradient launches; the main stream joins before returning.  All segments: +1.3 % clips/s
        # (round 1: 70.2 vs 71.2 ms / step; round 4: 829.8 vs 819.3 clips/s).  Default "1": every segment.  "auto": every segment EXCEPT
        # layer1 (rounds 1-3's default: with two kernels sharing the CUs a per-launch duration stops describing one kernel, and
        # bench.py's roofline object was defined on all of layer1's launches; it now times layer1's FORWARD launches, which no
        # side-stream kernel overlaps, and reports the overlapped data-gradient launches beside them).  "0": off.
        mode = os.environ.get("SLIC_WGRAD_STREAM", "1")
        side = self._side_stream() if (mode == "1" or (mode == "auto" and si != 1)) else None
        main = torch.cuda.current_stream()

        def wgrad_async(plan, x, dz, weight):
            dW = new_like(weight)                       # allocated on the main stream: it outlives the side stream's use
            if side is None:
                return plan.wgrad(x, dz, B, dW)
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            for t in (x, dz, dW):
                t.record_stream(side)                   # the caching allocator must not recycle them under the side kernel
            with torch.cuda.stream(side):
                plan.wgrad(x, dz, B, dW)
            return dW

        def join():
            if side is not None:
                ev = torch.cuda.Event()
                ev.record(side)
                main.wait_event(ev)

        if si == 5:
            dy = dout
            if net.projection_head:
                ah, h1, bnp, pooled = ctx["ah"], ctx["h1"], ctx["bnp"], ctx["pooled"]
                d5 = dy.view(B, 1, 1, 1, -1)
                grads[net.fc2.weight] = self.fc2.wgrad(ah.view(B, 1, 1, 1, -1), d5, B, new_like(net.fc2.weight))
                grads[net.fc2.bias] = bias_grad(dy, net.fc2.bias)
                dah = self.fc2.dgrad(d5, self.fc2.pack_dgrad(net.fc2.weight), B)
                dh1, _, dg, db = self._bn_bwd(dah.view(B, -1), ah.view(B, -1), h1.view(B, -1), bnp, False)
                grads[net.bn_proj.weight], grads[net.bn_proj.bias] = dg, db
                grads[net.fc1.weight] = self.fc1.wgrad(pooled.view(B, 1, 1, 1, -1), dh1.view(B, 1, 1, 1, -1), B, new_like(net.fc1.weight))
                grads[net.fc1.bias] = bias_grad(dh1.view(B, -1), net.fc1.bias)
                dpool = self.fc1.dgrad(dh1.view(B, 1, 1, 1, -1), self.fc1.pack_dgrad(net.fc1.weight), B).view(B, -1)
            else:
                dpool = dy
            To, Ho, Wo = self.final_dims
            S = To * Ho * Wo
            dx = torch.empty(ctx["last_shape"], dtype=torch.float32, device=dout.device)
            call("slic_avgpool_bwd", ptr(dpool), B, S, self.feat, ptr(dx), stream())
            return dx, grads
        if si >= 1:
            blocks = list(zip(self.layer_blocks[si - 1], ctx["blocks"]))
            fuse = os.environ.get("SLIC_BN_FUSE", "1") != "0"
            pre = None          # (g, partial) when the previous dgrad already produced this block's masked gradient + sums
            pf = self._prefused.pop((pid, si), None)
            if pf is not None and pf[0] == dout.data_ptr() and pf[1] == tuple(dout.shape):
                pre = (dout, pf[2])
            for bi in reversed(range(len(blocks))):
                (blk, p1, p2, pd), s = blocks[bi]
                p3 = self.p3.get(blk)
                # out = relu(bn_l(conv_l(.)) + r), l = the block's last convolution (conv2; conv3 of a Bottleneck)
                if pre is not None:
                    g, part = pre
                    dzl, dgl, dbl = self._bn_bwd_fused(part, g, s["zl"], s["bl"])
                    pre = None
                else:
                    dzl, g, dgl, dbl = self._bn_bwd(dout, s["out"], s["zl"], s["bl"], True)
                if p3 is not None:
                    grads[blk.bn3.weight], grads[blk.bn3.bias] = dgl, dbl
                    grads[blk.conv3.weight] = wgrad_async(p3, s["a2"], dzl, blk.conv3.weight)
                    # a2 = relu(bn2(conv2(a1))): mask + BatchNorm-backward sums on conv3's dgrad epilogue
                    b2 = s["b2"]
                    if fuse:
                        g2, part2 = p3.dgrad(dzl, p3.pack_dgrad(blk.conv3.weight), B, mask=s["a2"], bwd=(s["z2"], b2.mean, b2.invstd))
                        del dzl
                        dz2, dg2, db2 = self._bn_bwd_fused(part2, g2, s["z2"], b2)
                        del g2
                    else:
                        da2 = p3.dgrad(dzl, p3.pack_dgrad(blk.conv3.weight), B)
                        del dzl
                        dz2, _, dg2, db2 = self._bn_bwd(da2, s["a2"], s["z2"], b2, False)
                        del da2
                else:
                    dz2, dg2, db2 = dzl, dgl, dbl
                    del dzl
                grads[blk.bn2.weight], grads[blk.bn2.bias] = dg2, db2
                grads[blk.conv2.weight] = wgrad_async(p2, s["a1"], dz2, blk.conv2.weight)
                # a1 = relu(bn1(conv1(x))): the ReLU mask and the BatchNorm-backward sums ride on conv2's dgrad epilogue
                b1 = s["b1"]
                if fuse:
                    g1, part1 = p2.dgrad(dz2, p2.pack_dgrad(blk.conv2.weight), B, mask=s["a1"], bwd=(s["z1"], b1.mean, b1.invstd))
                    del dz2
                    dz1, dg1, db1 = self._bn_bwd_fused(part1, g1, s["z1"], b1)
                    del g1
                else:
                    da1 = p2.dgrad(dz2, p2.pack_dgrad(blk.conv2.weight), B)
                    del dz2
                    dz1, _, dg1, db1 = self._bn_bwd(da1, s["a1"], s["z1"], b1, False)
                    del da1
                grads[blk.bn1.weight], grads[blk.bn1.bias] = dg1, db1
                grads[blk.conv1.weight] = wgrad_async(p1, s["x"], dz1, blk.conv1.weight)
                # the layer below consumes dx through relu + BatchNorm (the previous block's bn2, or across the segment
                # boundary the previous segment's last bn2 / the stem's bn1): fuse its mask and sums as well
                below = None
                if fuse and bi > 0:
                    pb = blocks[bi - 1][1]
                    below = (pb["out"], pb["zl"], pb["bl"])
                elif fuse and bi == 0:
                    prev = self._live_ctx(pid, si - 1)
                    if prev is not None and si - 1 == 0:
                        if self.pool_in is None:           # with the max-pool between, dx does not reach the stem's ReLU directly
                            below = (prev["a0"], prev["z0"], prev["bn0"])
                    elif prev is not None:
                        pb = prev["blocks"][-1]
                        below = (pb["out"], pb["zl"], pb["bl"])
                kw = {}
                if below is not None:
                    kw = dict(mask=below[0], bwd=(below[1], below[2].mean, below[2].invstd))
                if pd is not None:
                    # r = bn_d(conv_d(x)): g is its upstream gradient
                    dzd, _, dgd, dbd = self._bn_bwd(g, None, s["zd"], s["bd"], False)
                    grads[blk.downsample[1].weight], grads[blk.downsample[1].bias] = dgd, dbd
                    grads[blk.downsample[0].weight] = wgrad_async(pd, s["x"], dzd, blk.downsample[0].weight)
                    # the shortcut's data gradient reaches one input position in stride^3 (1x1x1 kernel): only that parity class is
                    # launched (seven launches that wrote zeros at layer2.0 / 3.0 / 4.0 before), and conv1's data gradient takes dx as
                    # its addend on that class alone — with the same strides the classes of the two plans coincide
                    sparse = pd.stride == p1.stride and pd.stride != (1, 1, 1) and pd.tap_classes() <= p1.tap_classes()
                    dx = pd.dgrad(dzd, pd.pack_dgrad(blk.downsample[0].weight), B, skip_empty=sparse)
                    res = p1.dgrad(dz1, p1.pack_dgrad(blk.conv1.weight), B, addend=dx, out=dx,
                                   addend_classes=pd.tap_classes() if sparse else None, **kw)
                elif blk in self.short_a:
                    # the reference's shortcut 'A' concatenates `out.data` (models/resnet.py:220): no gradient through the branch
                    res = p1.dgrad(dz1, p1.pack_dgrad(blk.conv1.weight), B, **kw)
                else:
                    res = p1.dgrad(dz1, p1.pack_dgrad(blk.conv1.weight), B, addend=g, **kw)
                if below is not None and bi > 0:
                    pre = res
                    dout = None
                elif below is not None:
                    dout, part = res             # crosses the segment boundary through autograd: the sums travel beside it
                    self._prefused[(pid, si - 1)] = (dout.data_ptr(), tuple(dout.shape), part)
                else:
                    dout = res
            join()
            return dout, grads
        # stem: a0 = relu(bn1(conv1(x4))) (-> max-pool); the clip needs no gradient
        if self.pool_in is not None:
            T, H, W = self.pool_in
            da0 = torch.empty_like(ctx["a0"])
            call("slic_maxpool3d_bwd", ptr(dout), ptr(ctx["pool_arg"]), B, T, H, W, da0.shape[-1], ptr(da0), stream())
            dout = da0
        pf = self._prefused.pop((pid, 0), None)
        if pf is not None and pf[0] == dout.data_ptr() and pf[1] == tuple(dout.shape):
            dz0, dg0, db0 = self._bn_bwd_fused(pf[2], dout, ctx["z0"], ctx["bn0"])
        else:
            dz0, _, dg0, db0 = self._bn_bwd(dout, ctx["a0"], ctx["z0"], ctx["bn0"], False)
        grads[net.bn1.weight], grads[net.bn1.bias] = dg0, db0
        grads[net.conv1.weight] = wgrad_async(self.stem, ctx["x4"], dz0, net.conv1.weight)
        join()
        return None, grads

    # ------------------------------------------------------------------ whole passes (inference, tests, diagnostics)
    def forward(self, x, training, save):
        """x: [B, C, T, H, W] fp32 device tensor.  Returns (output [B, out_dim], [ctx per segment] or None)."""
        ctxs = []
        a = x
        self.prepack(with_dgrad=save)
        for si in range(self.N_SEG):
            a, c = self.seg_forward(si, a, training, save)
            ctxs.append(c)
        return a, (ctxs if save else None)

    def backward(self, ctxs, dy):
        """dy: [B, out_dim].  Returns {parameter: gradient} (reference layouts)."""
        grads = {}
        d = dy
        self._pass_id += 1
        pid = self._pass_id
        holders = [_Saved(c) for c in ctxs]                # kept alive for the duration of this call
        for si, hd in enumerate(holders):
            self._live[(pid, si)] = weakref.ref(hd)
        try:
            for si in reversed(range(self.N_SEG)):
                d, g = self.seg_backward(si, ctxs[si], d, pid)
                grads.update(g)
        finally:
            for si in range(self.N_SEG):
                self._live.pop((pid, si), None)
                self._prefused.pop((pid, si), None)
        return grads


class _Saved:
    """weak-referenceable holder of a segment's saved context (dicts are not)"""
    __slots__ = ("d", "__weakref__")

    def __init__(self, d):
        self.d = d


class _SegmentFn(torch.autograd.Function):
    """one segment of the encoder as an autograd node: forward/backward are the engine's hand-written passes"""

    @staticmethod
    def forward(ctx, inp, engine, si, training, *params):
        out, saved = engine.seg_forward(si, inp, training=training, save=True)
        # the saved context must not hold the output OBJECT (output -> grad_fn -> ctx -> output would be a reference
        # cycle that pins HBM until Python's cycle collector runs): keep a detached alias of the same storage
        if si == 0 and engine.pool_in is None:
            saved["a0"] = out.detach()
        elif 1 <= si <= 4:
            saved["blocks"][-1]["out"] = out.detach()
        ctx.engine, ctx.si, ctx.saved, ctx.params = engine, si, saved, params
        if si == 0:
            engine._pass_id += 1
            if len(engine._live) > 64:                     # drop entries of passes whose graphs were freed without a backward
                for key in [k for k, ref in engine._live.items() if ref() is None]:
                    engine._live.pop(key, None)
                    engine._prefused.pop(key, None)
        ctx.pid = engine._pass_id
        ctx.holder = _Saved(saved)    # the segment above fuses this segment's last ReLU/BatchNorm backward into its dgrad
        engine._live[(ctx.pid, si)] = weakref.ref(ctx.holder)
        ctx.inp_grad = inp.requires_grad
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.saved is None:
            raise RuntimeError("the encoder's saved activations are released by its first backward pass: a second backward "
                               "through the same forward (retain_graph=True) is not supported — run the forward again")
        dinp, grads = ctx.engine.seg_backward(ctx.si, ctx.saved, dout, ctx.pid)
        ctx.saved = None
        ctx.holder = None
        ctx.engine._live.pop((ctx.pid, ctx.si), None)
        return (dinp if ctx.inp_grad else None, None, None, None) + tuple(grads.get(p) for p in ctx.params)


class ResNet(nn.Module):
    """Same constructor, attributes and state_dict as the reference's ResNet (models/resnet.py:100-312)."""

    def __init__(self, block, layers, block_inplanes, n_input_channels=3, conv1_t_size=7, conv1_t_stride=1,
                 no_max_pool=False, shortcut_type='B', widen_factor=1.0, hidden_layer=2048, out_dim=128,
                 predict_temporal_ds=False, spatio_temporal_attention=False, projection_head=True,
                 num_classes=101, classifier=False, dropout=None):
        super().__init__()
        if spatio_temporal_attention or predict_temporal_ds or classifier:
            raise NotImplementedError("attention / temporal-ds / classifier heads are off in every SLIC config "
                                      "(config/default_params.py:97) and outside the hot path")
        block_inplanes = [int(x * widen_factor) for x in block_inplanes]
        self.in_planes = block_inplanes[0]
        self.no_max_pool = no_max_pool
        self.conv1 = nn.Conv3d(n_input_channels, self.in_planes, kernel_size=(conv1_t_size, 7, 7),
                               stride=(conv1_t_stride, 2, 2), padding=(conv1_t_size // 2, 3, 3), bias=False)
        self.bn1 = nn.BatchNorm3d(self.in_planes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool3d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, block_inplanes[0], layers[0], shortcut_type)
        self.layer2 = self._make_layer(block, block_inplanes[1], layers[1], shortcut_type, stride=2)
        self.layer3 = self._make_layer(block, block_inplanes[2], layers[2], shortcut_type, stride=2)
        self.layer4 = self._make_layer(block, block_inplanes[3], layers[3], shortcut_type, stride=2)
        self.spatio_temporal_attention = spatio_temporal_attention
        self.avgpool = nn.AdaptiveAvgPool3d((1, 1, 1))
        self.predict_temporal_ds = predict_temporal_ds
        self.projection_head = projection_head
        self.classifier = classifier
        self.num_classes = num_classes
        self.dropout = dropout
        if projection_head:
            print('==> setting up non-linear project heads')
            self.fc1 = nn.Linear(block_inplanes[3] * block.expansion, hidden_layer)
            self.bn_proj = nn.BatchNorm1d(hidden_layer)
            self.fc2 = nn.Linear(hidden_layer, out_dim)
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._engines = {}
        self.register_state_dict_pre_hook(_flush_engine_counters)

    def _make_layer(self, block, planes, blocks, shortcut_type, stride=1):
        downsample = None
        if stride != 1 or self.in_planes != planes * block.expansion:
            if shortcut_type == 'A':
                downsample = partial(self._downsample_basic_block, planes=planes * block.expansion, stride=stride)
            else:
                downsample = nn.Sequential(conv1x1x1(self.in_planes, planes * block.expansion, stride),
                                           nn.BatchNorm3d(planes * block.expansion))
        layers = [block(in_planes=self.in_planes, planes=planes, stride=stride, downsample=downsample)]
        self.in_planes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.in_planes, planes))
        return nn.Sequential(*layers)

    def _downsample_basic_block(self, x, planes, stride):
        """shortcut 'A' (models/resnet.py:213-222): every stride-th position, channels zero-padded to `planes`; the engine runs it
        as one kernel (slic_shortcut_a) from the keywords of the partial that _make_layer stores in the block"""
        raise RuntimeError("the shortcut runs inside ResNet.forward's HIP plan")

    def __getstate__(self):
        # copy.deepcopy(model) / torch.save(model): the execution plans (device tables, streams, events) are rebuilt on demand
        state = self.__dict__.copy()
        state["_engines"] = {}
        return state

    def _engine(self, x):
        key = (tuple(x.shape), str(x.device))
        eng = self._engines.get(key)
        if eng is None:
            eng = _Engine(self, x.shape, x.device)
            self._engines[key] = eng
        return eng

    def forward(self, x):
        if not x.is_cuda:
            raise _lib.SlicError("ResNet.forward needs a gfx950 device tensor: the encoder has no CPU/PyTorch fallback")
        _lib.load()
        x = x.to(torch.float32)
        return run_engine(self._engine(x), self, x)


def _flush_engine_counters(module, prefix, keep_vars):
    """state_dict pre-hook: the engines bump num_batches_tracked once per pass, at the head segment — a state_dict taken between two segments
    (or after an aborted pass) first gets the pending bumps"""
    for eng in getattr(module, "_engines", {}).values():
        flush = getattr(eng, "flush_counters", None)
        if flush is not None:
            flush()


def run_engine(eng, module, x):
    """drive one encoder through its engine: autograd segments whenever a backward may follow — train mode (batch statistics)
    or eval mode (BatchNorm frozen on its running statistics: the reference's ResNet.forward is an ordinary autograd graph in
    any mode, models/resnet.py:255-312) — a plain inference pass with BatchNorm folded into the conv epilogues otherwise"""
    params = [p for p in module.parameters() if p.requires_grad]
    eng.grad_views = getattr(module, "_slic_grad_views", None)
    if torch.is_grad_enabled() and params:
        a = x
        eng.prepack(with_dgrad=True)
        for si in range(eng.N_SEG):
            a = _SegmentFn.apply(a, eng, si, bool(module.training), *eng.seg_params(si))
        return a
    if torch.is_grad_enabled() and x.requires_grad:
        # a fully frozen encoder under an input that wants a gradient: the reference would back-propagate to the clip
        # (models/resnet.py:255-312 is an ordinary autograd graph); the stem here has no data gradient — say so instead of
        # handing back a graph-less tensor
        raise _lib.SlicError("ResNet.forward: the input requires a gradient but no encoder parameter does; the stem has no data "
                             "gradient on this path (detach the clip, or leave at least one parameter trainable)")
    with torch.no_grad():
        return eng.forward(x, training=module.training, save=False)[0]


def generate_model(model_depth, **kwargs):
    """models/resnet.py:436-456"""
    def get_inplanes():
        return [64, 128, 256, 512]

    assert model_depth in [10, 18, 34, 50, 101, 152, 200]
    if model_depth == 10:
        return ResNet(BasicBlock, [1, 1, 1, 1], get_inplanes(), **kwargs)
    if model_depth == 18:
        return ResNet(BasicBlock, [2, 2, 2, 2], get_inplanes(), **kwargs)
    if model_depth == 34:
        return ResNet(BasicBlock, [3, 4, 6, 3], get_inplanes(), **kwargs)
    if model_depth == 50:
        return ResNet(Bottleneck, [3, 4, 6, 3], get_inplanes(), **kwargs)
    if model_depth == 101:
        return ResNet(Bottleneck, [3, 4, 23, 3], get_inplanes(), **kwargs)
    if model_depth == 152:
        return ResNet(Bottleneck, [3, 8, 36, 3], get_inplanes(), **kwargs)
    return ResNet(Bottleneck, [3, 24, 36, 3], get_inplanes(), **kwargs)
