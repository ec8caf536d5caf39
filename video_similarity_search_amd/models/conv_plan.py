"""
Host-side plans for the table-driven gather-GEMM (csrc/conv.hip): builds the K-chunk tables for
forward / data-gradient launches of one 3-D convolution (or linear layer) and marshals
`SlicConvArgs`.  Stands in for what cuDNN's descriptors/algorithm search do behind nn.Conv3d in the
reference (models/resnet.py:11-25,126-131; cudnn.benchmark, online_train.py:444).

Layout contract: activations NDHWC fp32 contiguous [B, T, H, W, C] with C % 4 == 0; weights stay in
the reference layout [N, C, kt, kh, kw] (state_dict-compatible) and are re-packed on the device.
"""
import ctypes
import itertools
import os

import numpy as np
import torch

from .. import _lib
from .._lib import SlicConvArgs, call, ptr, stream


def _pad8(n):
    return (n + 7) // 8 * 8


def _pack_off(oa, ob, oc):
    assert -128 <= oa < 128 and -128 <= ob < 128 and -128 <= oc < 128
    return (oa + 128) | ((ob + 128) << 8) | ((oc + 128) << 16)


_ENV_KEYS = {}


def _env_fast_path():
    """os.environ's private byte dictionary, or None where it cannot be trusted: `_data` is a CPython implementation detail (bytes keys on
    POSIX), so the fast path is taken only on POSIX, only when `_data` is a real dict, and only after a self-check against os.environ.get on
    a key that is set AND one that is not — anywhere else every switch goes through os.environ.get (a different
    implementation would otherwise read every switch, the launch limits included, as unset)."""
    data = getattr(os.environ, "_data", None)
    if os.name != "posix" or not isinstance(data, dict) or not hasattr(os.environ, "encodekey") or not hasattr(os.environ, "decodevalue"):
        return None
    try:
        probe = next(iter(os.environ), None)
        if probe is not None and os.environ.decodevalue(data[os.environ.encodekey(probe)]) != os.environ.get(probe):
            return None
        if data.get(os.environ.encodekey("SLIC__NEVER_SET__PROBE")) is not None:
            return None
    except Exception:
        return None
    return data


_ENV_DATA = _env_fast_path()


def _env(name, default=None):
    """os.environ.get without its exception path: the launch rules below read a dozen switches in front of every launch (350 reads per
    training step), almost all unset — Mapping.get raises and catches a KeyError for each of those (~1.2 us); a lookup in the
    environment's own byte dictionary does not.  Sees monkeypatched / late-set variables like os.environ does."""
    data = _ENV_DATA
    if data is None:
        return os.environ.get(name, default)
    key = _ENV_KEYS.get(name)
    if key is None:
        key = _ENV_KEYS[name] = os.environ.encodekey(name)
    v = data.get(key)
    return default if v is None else os.environ.decodevalue(v)


def _tap_mask(oa, ob, oc):
    """bit (7*dim + o + 3): the row's precomputed in-bounds mask must contain all three"""
    assert max(abs(oa), abs(ob), abs(oc)) <= 3, "tap offsets beyond +-3 are not supported by the row mask"
    return (1 << (oa + 3)) | (1 << (7 + ob + 3)) | (1 << (14 + oc + 3))


class ConvPlan:
    """One conv layer at one input size.  kernel/stride/pad are (t, h, w) triples."""

    WINO_MIN_WGS = 384     # forward / data gradient: a launch of fewer 64-tile x 64-n workgroups than this (layer4 at B = 32: 112) cuts its K loop

    WINO2_PIECES = (2, 3, 4, 6, 8, 12, 16)     # K-split pieces of a variant-31 tail (measured at layer4 / layer2 and at small batches, scripts/r4/pieces.sh)
    WINO2_MIN_WGS = 48     # two-dimensional Winograd (variant 31): launches of fewer 64-tile x 64-n workgroups stay on variant 30 (launches of less than
                           # a dispatch round of the 256 one-per-CU slots cut their K loop into pieces: _plan_split).  Measured at B = 8, algorithmic
                           # TFLOP/s forward / data gradient, 2-D vs 1-D: layer3 (56 workgroups x 4 pieces) 178 / 182 vs 140 / 144; layer4 (16
                           # workgroups, best at 4 pieces) 53 / 54 vs 60 / 62 (scripts/r4/small_batch.sh)

    def __init__(self, C, N, kernel, stride, pad, in_dims, device, wrun=None, wino=None, wino2=None, batch=None, wino2_wgrad=None):
        self.C, self.N = int(C), int(N)
        self.Cs = (self.C + 3) // 4 * 4
        self.kernel, self.stride, self.pad = tuple(kernel), tuple(stride), tuple(pad)
        self.in_dims = tuple(int(v) for v in in_dims)
        self.out_dims = tuple((i + 2 * p - k) // s + 1 for i, p, k, s in zip(self.in_dims, self.pad, self.kernel, self.stride))
        self.ntaps = int(np.prod(self.kernel))
        self.device = device
        assert self.N % 4 == 0, "output channels must be a multiple of 4"
        # W-run operand (the RGB stem): with C = 3 padded to 4 channels a third of the K dimension is zeros (7^3 x 4 = 1372 for
        # 1029 real MACs per output).  Instead keep the clip as [B, T, H, W + 2 pad_w, C] (zero columns in W, C un-padded): the
        # kw taps of one (kt, kh) row are then ONE contiguous run of kw * C floats (21 -> padded to 24 with zero weights), in
        # bounds along W by construction; K = kt * kh * 24 = 1176 (12.5 % padding instead of 33 %).
        self.wrun = (self.C % 4 != 0 and self.kernel[2] * self.C <= 64) if wrun is None else bool(wrun)
        # Winograd F(4, 3) along W (variant 30 of slic_conv_gemm): the 3 x 3 x 3 stride-1 pad-1 layers with 64-multiple channel
        # counts on both sides (forward reduces over C, the data gradient over N) — layers 1-4 of R3D-18.  Exact fp32, half the
        # multiplies.  wino=None: on where eligible unless SLIC_WINO=0; explicit variants of forward() / dgrad() need a plan built
        # with wino=False (the packed operand differs).
        base = (self.kernel == (3, 3, 3) and self.stride == (1, 1, 1) and self.pad == (1, 1, 1) and self.C % 64 == 0 and
                self.N % 64 == 0 and not self.wrun)
        Wd = self.in_dims[2]
        Wp = (Wd + 3) // 4 * 4
        # forward / data gradient: a width that is not a multiple of 4 runs with a ragged last tile per row (padded width must
        # divide 128: 14 -> 16, 7 -> 8); a launch of few workgroups cuts its K loop across workgroups (_plan_split)
        # and the kernel's K loop addresses 8-channel stages with a shift: C / 8 (forward) and N / 8 (data gradient, whose source
        # channels are N) must be powers of two — widths such as 192 or 384 (RESNET.WIDEN_FACTOR 1.5 / 3) stay on the direct kernels
        pow2 = lambda v: v > 0 and (v & (v - 1)) == 0
        eligible = base and (Wd % 4 == 0 or 128 % Wp == 0) and pow2(self.C // 8) and pow2(self.N // 8)
        on = _env("SLIC_WINO", "1") != "0"
        self.wino = (eligible and on) if wino is None else bool(wino)
        assert eligible or not self.wino, "Winograd F(4,3): 3x3x3 / stride 1 / pad 1, C and N 64 x a power of two, W % 4 == 0 or 4 ceil(W/4) | 128"
        # Winograd in two dimensions, F(4, 3) along W x F(2, 3) along H (variant 31, csrc/conv_wino2.hip): a third of the direct form's
        # multiplies instead of half.  Forward and data gradient of a Winograd plan whose blocks of 64 tiles (2 x 4 outputs each) all
        # hold the same number of real outputs (H even and W % 4 == 0; or H even and ceil(W / 4) | 64; or ceil(H / 2) ceil(W / 4) | 64)
        # and whose launch fills the chip: wino2=None decides from `batch` (the engine passes it), SLIC_WINO2=0 switches it off.
        Hd = self.in_dims[1]
        Hq2, Wq2 = (Hd + 1) // 2, (Wd + 3) // 4
        uniform = (Wd % 4 == 0 and Hd % 2 == 0) or (Hd % 2 == 0 and 64 % Wq2 == 0) or (64 % (Hq2 * Wq2) == 0)
        elig2 = eligible and uniform
        # size limits of the two-dimensional kernels (csrc/conv_wino2.hip: wino2_check, slic_conv_wgrad_wino2): 32-bit buffer offsets with
        # the out-of-range offset 0xFFFFFF00 standing for padding pixels — the source plus a frame on either side must stay below it
        # (layer1 at 112 x 112: B <= 331), and the weight gradient's tile records hold 24-bit positions.  A batch beyond the limits of
        # ONE launch is run in chunks of whole clips (_launch_batch; forward / dgrad / wgrad loop over them), so the gates below see
        # the clips of one launch, not the whole batch.
        self._base333 = bool(base)
        if batch is not None:
            batch = min(int(batch), self._launch_batch(int(batch)))
        positions = 0 if batch is None else int(batch) * int(np.prod(self.in_dims))
        fits2 = positions * max(self.C, self.N) * 4 + 2 * Hd * Wd * max(self.C, self.N) * 4 + 16 <= 0xFFFFFF00
        self.wino2_size_ok = bool(fits2)
        if wino2 is None:
            wgs = 0 if batch is None else -(-(int(batch) * self.in_dims[0] * Hq2 * Wq2) // 64) * (max(self.C, self.N) // 64)
            wino2 = (elig2 and self.wino and fits2 and _env("SLIC_WINO2", "1") != "0" and
                     wgs >= int(_env("SLIC_WINO2_MIN_WGS", self.WINO2_MIN_WGS)))
        self.wino2 = bool(wino2)
        assert (elig2 and self.wino) or not self.wino2, "Winograd F(4,3) x F(2,3): a Winograd plan with uniform 64-tile blocks"
        # weight gradient by the transposed algorithm: any width (its work splits over taps, channel blocks and tile slices)
        self.wino_wgrad = (base and on and _env("SLIC_WINO_WGRAD", "1") != "0") if wino is None else (bool(wino) and base)
        # ... and by the transposed TWO-dimensional algorithm (slic_conv_wgrad_wino2) wherever the two-dimensional forward runs and its
        # 3 x C / 64 x N / 64 workgroups per tile slice fit the 256 one-workgroup-per-CU slots (all four layers at B = 32; layer4: 192
        # workgroups, one slice).  Algorithmic TFLOP/s at B = 32, 2-D vs 1-D kernel: layer1 303 vs 230, layer2 289 vs 230, layer3 228 vs
        # 192, layer4 150 vs 143.  It takes any H and W (ragged tiles are masked), so it also runs where the forward stays one-dimensional
        # for want of workgroups — layer4 at B = 8: 94 vs 81 — as long as the launch has 128 tiles.
        # SLIC_WINO2_WGRAD=0 switches it off, =2 restricts it to the 128-channel layers.
        mode = _env("SLIC_WINO2_WGRAD", "1")
        blocks2 = 3 * (self.C // 64) * (self.N // 64)
        if wino2_wgrad is None:
            tiles2 = 0 if batch is None else int(batch) * self.in_dims[0] * Hq2 * Wq2
            wino2_wgrad = ((self.wino2 or (self.wino and tiles2 >= 128 and _env("SLIC_WINO2", "1") != "0")) and self.wino_wgrad and
                           fits2 and positions < (1 << 24) and mode != "0" and blocks2 <= int(_env("SLIC_WINO2_WGRAD_MAXBLOCKS", "256")) and
                           (mode != "2" or (self.C == 128 and self.N == 128)))
        self.wino2_wgrad = bool(wino2_wgrad)
        assert not self.wino2_wgrad or base, "transposed 2-D Winograd weight gradient: a Winograd plan"
        self._wu = self._wud = None
        self._wino_tabs = {}
        self._wino2_tabs = {}
        if self.wrun:
            self._init_wrun()
            return
        self.run_len = self.run_px = 0
        self.src_dims = self.in_dims
        # ---- forward table: k = tap*Cs + c
        T, H, W = self.in_dims
        kt, kh, kw = self.kernel
        nch = self.ntaps * self.Cs // 4
        self.nchunks_fwd = _pad8(nch)
        tab = np.zeros((self.nchunks_fwd, 4), np.int32)
        tab[:, 1] = -1
        q = np.arange(nch)
        tap, c = (q * 4) // self.Cs, (q * 4) % self.Cs
        dt, dh, dw = tap // (kh * kw), (tap // kw) % kh, tap % kw
        oa, ob, oc = dt - self.pad[0], dh - self.pad[1], dw - self.pad[2]
        tab[:nch, 0] = ((oa * H + ob) * W + oc) * self.Cs + c
        assert max(self.pad) <= 3 and all(k - 1 - p <= 3 for k, p in zip(self.kernel, self.pad))
        tab[:nch, 1] = (1 << (oa + 3)) | (1 << (7 + ob + 3)) | (1 << (14 + oc + 3))
        tab[:nch, 2] = q * 4
        tab[:nch, 3] = (oa + 128) | ((ob + 128) << 8) | ((oc + 128) << 16)
        self.tab_fwd = torch.from_numpy(tab).to(device)
        # per-TAP records for the LDS-DMA kernel (source channels % 32 == 0: a 32-wide K tile never straddles a tap)
        self.tap_fwd = None
        if self.Cs % 32 == 0 and self.ntaps <= 64:
            tt = np.arange(self.ntaps)
            tdt, tdh, tdw = tt // (kh * kw), (tt // kw) % kh, tt % kw
            ta, tb, tc = tdt - self.pad[0], tdh - self.pad[1], tdw - self.pad[2]
            trec = np.zeros((self.ntaps, 4), np.int32)
            trec[:, 0] = ((ta * H + tb) * W + tc) * self.Cs
            trec[:, 1] = (1 << (ta + 3)) | (1 << (7 + tb + 3)) | (1 << (14 + tc + 3))
            trec[:, 2] = tt * self.Cs
            self.tap_fwd = torch.from_numpy(trec).to(device)
        self.Kp = self.nchunks_fwd * 4                      # packed forward weight row length
        # ---- data-gradient tables: src = dY [B, To, Ho, Wo, N], k = tap*N + n, one per parity class
        self.Kd = _pad8(self.ntaps * self.N // 4) * 4
        To, Ho, Wo = self.out_dims
        self.dgrad_classes = []
        nq = self.N // 4
        for cls in itertools.product(*[range(s) for s in self.stride]):
            grid = tuple((d - p + s - 1) // s for d, p, s in zip(self.in_dims, cls, self.stride))
            if min(grid) <= 0:
                continue
            valid = []
            for dim in range(3):
                v = []
                for d in range(self.kernel[dim]):
                    num = cls[dim] + self.pad[dim] - d
                    if num % self.stride[dim] == 0:
                        v.append((d, num // self.stride[dim]))
                valid.append(v)
            rows, taprec = [], []
            for (dt_, oa_), (dh_, ob_), (dw_, oc_) in itertools.product(*valid):
                tap_ = (dt_ * kh + dh_) * kw + dw_
                base = ((oa_ * Ho + ob_) * Wo + oc_) * self.N
                po, tmk = _pack_off(oa_, ob_, oc_), _tap_mask(oa_, ob_, oc_)
                taprec.append((base, tmk, tap_ * self.N, 0))
                for n4 in range(nq):
                    rows.append((base + n4 * 4, tmk, tap_ * self.N + n4 * 4, po))
            nchd = _pad8(len(rows))
            t = np.zeros((max(nchd, 8), 4), np.int32)
            t[:, 1] = -1
            if rows:
                t[: len(rows)] = np.asarray(rows, np.int32)
            tapt = None
            if self.N % 32 == 0 and 0 < len(taprec) <= 64:
                tapt = torch.from_numpy(np.asarray(taprec, np.int32)).to(device)
            self.dgrad_classes.append(dict(cls=cls, grid=grid, nchunks=nchd if rows else 0,
                                           tab=torch.from_numpy(t).to(device), tap=tapt))
        self._wp = None
        self._wd = None
        self._wp_key = self._wd_key = None
        self._row_tabs = {}
        self.prof = None      # bench.py: list collecting (start, end) HIP event pairs around conv_gemm launches

    def _init_wrun(self):
        T, H, W = self.in_dims
        kt, kh, kw = self.kernel
        self.Cs = self.C                                     # channels per pixel of the source tensor: not padded
        self.run_px = kw
        self.run_len = (kw * self.C + 3) // 4 * 4
        # padded width: the conv's own zero columns, and enough on the right for the zero-weight tail of the last run of a row
        need = (self.out_dims[2] - 1) * self.stride[2] * self.C + self.run_len
        Wp = max(W + 2 * self.pad[2], (need + self.C - 1) // self.C)
        self.src_dims = (T, H, Wp)                           # what the kernel sees: its w coordinate is already shifted by pad_w
        cpr = self.run_len // 4                              # 16-byte chunks per run
        nch = kt * kh * cpr
        self.nchunks_fwd = _pad8(nch)
        tab = np.zeros((self.nchunks_fwd, 4), np.int32)
        tab[:, 1] = -1
        q = np.arange(nch)
        run, j = q // cpr, q % cpr
        oa, ob = run // kh - self.pad[0], run % kh - self.pad[1]
        assert max(self.pad[:2]) <= 3 and all(k - 1 - p <= 3 for k, p in zip(self.kernel[:2], self.pad[:2]))
        tab[:nch, 0] = ((oa * H + ob) * Wp) * self.C + 4 * j
        tab[:nch, 1] = (1 << (oa + 3)) | (1 << (7 + ob + 3)) | (1 << (14 + 0 + 3))
        tab[:nch, 2] = q * 4
        tab[:nch, 3] = (oa + 128) | ((ob + 128) << 8) | (128 << 16)
        self.tab_fwd = torch.from_numpy(tab).to(self.device)
        self.tap_fwd = None
        self.Kp = self.nchunks_fwd * 4
        self.dgrad_classes = []                              # the clip needs no gradient
        self._wp = self._wd = None
        self._wp_key = self._wd_key = None
        self._row_tabs = {}
        self.prof = None

    def make_source(self, x):
        """the operand layout of this plan from an NCDHW clip batch [B, C, T, H, W]"""
        B, C, T, H, W = x.shape
        assert C == self.C and (T, H, W) == self.in_dims
        x = x.contiguous()
        if self.wrun:
            y = torch.empty((B,) + self.src_dims + (self.C,), dtype=torch.float32, device=x.device)
            call("slic_ncdhw_to_ndhwc_wpad", ptr(x), B, C, T * H, W, self.pad[2], self.src_dims[2], ptr(y), stream())
        else:
            y = torch.empty(B, T, H, W, self.Cs, dtype=torch.float32, device=x.device)
            call("slic_ncdhw_to_ndhwc", ptr(x), B, C, T * H * W, self.Cs, ptr(y), stream())
        return y

    # ------------------------------------------------------------------ weights
    # The packed operands are kept per plan and reused while `weight` is the tensor they were packed from: the encoder packs
    # every layer's weights on a side stream when a pass starts (resnet._Engine.prepack — off the critical path), and the
    # launches below then find them ready.  The key is dropped at every pass entry, so a pack never outlives a pass.
    # CONTRACT for direct users of a ConvPlan: the key is (data_ptr, _version) of the weight tensor, and writes made through
    # `weight.data` bump a different version counter — call drop_packs() (or pass fresh=True) after ANY update of the weights,
    # otherwise pack_fwd / pack_dgrad hand back the operand packed from the old values.
    @staticmethod
    def _wkey(weight):
        return (weight.data_ptr(), weight._version)

    def drop_packs(self):
        self._wp_key = self._wd_key = None

    def pack_fwd(self, weight, fresh=False):
        if self.wino:
            if not fresh and self._wp_key is not None and self._wp_key == self._wkey(weight):
                return self._wu
            if self._wu is None:
                self._wu = torch.empty((72 if self.wino2 else 54) * self.C * self.N, dtype=torch.float32, device=self.device)
            call("slic_pack_weight_wino2" if self.wino2 else "slic_pack_weight_wino", ptr(weight), self.N, self.C, 0, ptr(self._wu), stream())
            self._wp_key = self._wkey(weight)
            return self._wu
        if not fresh and self._wp_key is not None and self._wp_key == self._wkey(weight):
            return self._wp
        if self._wp is None:     # zeroed once: the packer writes real elements only, the padding stays zero
            self._wp = torch.zeros(self.N, self.Kp, dtype=torch.float32, device=self.device)
        if self.wrun:
            call("slic_pack_weight_fwd_runs", ptr(weight), self.N, self.C, self.ntaps, self.run_len, self.run_px, self.Kp,
                 ptr(self._wp), stream())
        else:
            call("slic_pack_weight_fwd", ptr(weight), self.N, self.C, self.ntaps, self.Cs, self.Kp, ptr(self._wp), stream())
        self._wp_key = self._wkey(weight)
        return self._wp

    def pack_dgrad(self, weight, fresh=False):
        if self.wrun:
            raise _lib.SlicError("the W-run operand serves forward and weight gradient only (the clip needs no gradient)")
        if self.wino:
            if not fresh and self._wd_key is not None and self._wd_key == self._wkey(weight):
                return self._wud
            if self._wud is None:
                self._wud = torch.empty((72 if self.wino2 else 54) * self.C * self.N, dtype=torch.float32, device=self.device)
            call("slic_pack_weight_wino2" if self.wino2 else "slic_pack_weight_wino", ptr(weight), self.N, self.C, 1, ptr(self._wud), stream())
            self._wd_key = self._wkey(weight)
            return self._wud
        if not fresh and self._wd_key is not None and self._wd_key == self._wkey(weight):
            return self._wd
        if self._wd is None:
            self._wd = torch.zeros(self.Cs, self.Kd, dtype=torch.float32, device=self.device)
        call("slic_pack_weight_dgrad", ptr(weight), self.N, self.C, self.ntaps, self.Cs, self.Kd, ptr(self._wd), stream())
        self._wd_key = self._wkey(weight)
        return self._wd

    # ------------------------------------------------------------------ launches
    def _fwd_args(self, x, B):
        a = SlicConvArgs()
        T, H, W = self.src_dims
        To, Ho, Wo = self.out_dims
        assert tuple(x.shape) == (B, T, H, W, self.Cs) and x.is_contiguous(), (tuple(x.shape), (B, T, H, W, self.Cs))
        a.src = x.data_ptr()
        a.k_run_len, a.k_run_px = self.run_len, self.run_px
        a.src_bytes = _lib.u32_bytes(x, 'conv source')
        a.tab = self.tab_fwd.data_ptr()
        a.tap_tab = self.tap_fwd.data_ptr() if self.tap_fwd is not None else None
        a.M = B * To * Ho * Wo
        a.N = self.N
        a.nchunks = self.nchunks_fwd
        a.Cs, a.Ts, a.Hs, a.Ws = self.Cs, T, H, W
        a.Ga, a.Gb, a.Gc = To, Ho, Wo
        a.sa, a.sb, a.sc = self.stride
        a.ldw, a.ldo = self.Kp, self.N
        return a

    # ---- batches beyond one launch's range ------------------------------------------------------------------------------------------
    # The kernels address their operands with 32-bit byte offsets (buffer resources: out-of-range lanes read zeros, which is what makes
    # padding taps and ragged tiles branch-free) and the two-dimensional Winograd weight gradient keeps 24-bit positions in its tile
    # records.  A tensor of one launch must therefore stay below 4 GiB - 4 KiB (less two frames for variant 31, whose resource starts a
    # frame early) and, for the 3 x 3 x 3 stride-1 layers, below 2^24 positions: layer1 at 112 x 112 reaches that at 331 clips, at 128 x 128
    # at 253.  The reference's cuDNN path has no such limit (train-mode BatchNorm needs the whole batch, so the CALLER cannot split), so a
    # larger batch is run in chunks of whole clips inside the plan: convolution is independent per clip, the BatchNorm statistic slabs of
    # the chunks concatenate (a chunk is a whole number of slab rows), the weight gradients of the chunks are added in chunk order.
    # SLIC_CONV_MAX_BYTES / SLIC_CONV_MAX_POSITIONS lower the limits (tests force chunking at small sizes).
    def _launch_batch(self, B):
        # (called in front of every launch: the answer is cached per batch size and limit setting — 43 calls per backward at ~16 us each
        #  were 0.7 ms of a 6.8 ms host step before the cache, scripts/r5/host_parts.py)
        key = (B, _env("SLIC_CONV_MAX_BYTES"), _env("SLIC_CONV_MAX_POSITIONS"), hasattr(self, "src_dims"))
        cache = self.__dict__.setdefault("_lb_cache", {})
        hit = cache.get(key)
        if hit is not None:
            return hit
        lim_b = int(key[1]) if key[1] is not None else 0xFFFFFF00 - 4096
        lim_p = int(key[2]) if key[2] is not None else (((1 << 24) - 1) if self._base333 else ((1 << 31) - 1))
        src_dims = getattr(self, "src_dims", self.in_dims)
        pin, pout = int(np.prod(src_dims)), int(np.prod(self.out_dims))
        cmax = max(self.Cs, self.N)
        per_clip = max(pin, int(np.prod(self.in_dims)), pout) * cmax * 4
        margin = 2 * self.in_dims[1] * self.in_dims[2] * cmax * 4 + 256 if self._base333 else 0
        Bc = min(B, max(0, lim_b - margin) // per_clip, lim_p // max(pin, pout))
        if Bc < B:
            Bc = Bc // 8 * 8                   # whole slab rows per chunk at every block size in use (128 / 392 / 448 / 512 rows)
            if Bc < 8:
                raise _lib.SlicError(f"ConvPlan: eight clips of {self.in_dims} x {cmax} channels exceed one launch's 32-bit range")
        else:
            Bc = B
        cache[key] = Bc
        return Bc

    def _chunks(self, B):
        Bc = self._launch_batch(B)
        return None if Bc >= B else [(b0, min(b0 + Bc, B)) for b0 in range(0, B, Bc)]

    def forward(self, x, wp, B, bias=None, scale=None, shift=None, addend=None, relu=False, want_stats=False,
                variant=0, _out=None):
        """x: [B, T, H, W, Cs] -> z: [B, To, Ho, Wo, N]; returns (z, (stat_partial, rows_per_partial) or None)"""
        lib = _lib.load()
        chunks = self._chunks(B)
        if chunks is not None:
            z = torch.empty((B,) + self.out_dims + (self.N,), dtype=torch.float32, device=x.device)
            parts, tms = [], set()
            for b0, b1 in chunks:
                _, st = self.forward(x[b0:b1], wp, b1 - b0, bias, scale, shift, None if addend is None else addend[b0:b1], relu,
                                     want_stats, variant, _out=z[b0:b1])
                if want_stats:
                    parts.append(st[0])
                    tms.add(st[1])
                    assert b1 == B or ((b1 - b0) * int(np.prod(self.out_dims))) % st[1] == 0, "a chunk must be whole slab rows"
            if not want_stats:
                return z, None
            assert len(tms) == 1
            return z, (torch.cat(parts), tms.pop())
        self._prof_tag = "fwd"
        a = self._fwd_args(x, B)
        if self.wino:
            assert variant in (0, 30, 31) and bias is None, "a Winograd plan runs variant 30 / 31 only (build the plan with wino=False)"
            variant = 31 if self.wino2 else 30
        z = _out if _out is not None else torch.empty((B,) + self.out_dims + (self.N,), dtype=torch.float32, device=x.device)
        assert z.is_contiguous() and tuple(z.shape) == (B,) + self.out_dims + (self.N,)
        a.wgt = wp.data_ptr()
        a.wgt_bytes = _lib.u32_bytes(wp, 'packed weights')
        a.dst = z.data_ptr()
        a.bias = bias.data_ptr() if bias is not None else None
        a.scale = scale.data_ptr() if scale is not None else None
        a.shift = shift.data_ptr() if shift is not None else None
        a.addend = addend.data_ptr() if addend is not None else None
        a.relu = int(relu)
        part = None
        tm = lib.slic_conv_tile_m(ctypes.byref(a), self._pick(a, variant))
        if variant == 30 and self._plan_split(a, 30) is not None:
            tm = 128                         # the split-K finish pass works on blocks of 128 real GEMM rows
        if want_stats:
            R = (a.M + tm - 1) // tm
            part = torch.empty(R, 2, self.N, dtype=torch.float32, device=x.device)
            a.stat_partial = part.data_ptr()
        self._launch(a, variant)
        return z, ((part, tm) if want_stats else None)

    @staticmethod
    def _pick(a, variant):
        """variant 0 = auto: the LDS-DMA kernel wherever a per-tap table exists (source channels % 32 == 0) — 128 x 64 tiles
        for the tall N <= 64 layers (layer1: 52 % of the FLOPs), 64 x 64 tiles (5 workgroups / CU) otherwise — else the
        register-staged kernel (W-run stem, tiny-channel layers).  Measured with scripts/bench_conv.py."""
        if variant in (30, 31):
            return variant
        if not a.tap_tab:
            return 0
        if variant == 0:
            return 22 if (a.M >= 100000 and a.N <= 64) else 20
        return variant

    SPLIT_MIN_KTILES = 16      # k-tiles of 32 a split piece keeps at least
    SLOTS = {20: 256 * 5, 22: 256 * 3}     # workgroups the chip holds at once (CUs x workgroups per CU the LDS ring allows)

    @classmethod
    def _plan_split(cls, a, variant):
        """(nfull_rb, splits) for slic_conv_gemm_tailsplit, or None for a plain launch.

        A launch of T tiles on a chip that holds `slots` workgroups runs as T / slots rounds, and its last, partly filled
        round costs a whole round's time at a fraction of the chip: layer3 at B = 32 has 1568 64 x 64 tiles for 1280 slots,
        layer4 392.  So the tiles of the full rounds run whole, and the remainder — whole row blocks at the end of M — is cut
        along K into as many pieces as fill the slots once more (each piece keeps >= SPLIT_MIN_KTILES k-tiles).  A remainder
        that already fills most of a round is left alone.  SLIC_CONV_TAIL=0 switches the mechanism off (tests compare)."""
        if variant == 31:
            # two-dimensional Winograd, ONE workgroup per CU (256 slots): a launch's partly filled last dispatch round — when it is at
            # most half full — and launches of less than a round cut their K loop into as many even pieces as fill the slots once
            # (pieces | 3 Cs / 16; + a finish pass): layer2 at B = 32 is 784 workgroups = 3.06 rounds (16 in the tail), layer4 64 x 4 pieces
            if _env("SLIC_WINO2_SPLIT", "1") == "0":
                return None
            H2, W2 = a.Hs, a.Ws
            gx, ny = -(-((a.M // (H2 * W2)) * ((H2 + 1) // 2) * ((W2 + 3) // 4)) // 64), a.N // 64
            wgs = gx * ny
            rem = wgs % 256
            if rem == 0 or rem > 128 or wgs > int(_env("SLIC_WINO2_TAIL_MAXROUNDS", "6")) * 256:       # launches of many rounds are left alone (layer1: 12.25 — the tail's pieces + finish
                return None                                  # pass measured no faster there)
            # round 6: a launch the PERSISTENT kernel takes (whole 2 x 4 tiles everywhere, >= 2 blocks per compute unit) ends in column-half items
            # instead — two workgroups share a left-over block, no slab, no finish pass (SLIC_WINO2_HALFTAIL=0: the K-split tail as before)
            if (_env("SLIC_WINO2_HALFTAIL", "1") != "0" and _env("SLIC_WINO2_PERSIST", "1") != "0" and H2 % 2 == 0 and W2 % 4 == 0
                    and wgs >= 512 and not _env("SLIC_WINO2_PERSIST_GRID")):
                return None
            tail_x = -(-rem // ny)
            forced = int(_env("SLIC_WINO2_PIECES", "0"))
            units = 3 * a.Cs // 16
            # a piece keeps >= 48 of the 3 Cs / 4 stages when the whole launch is cut (layer3 at B = 8: 4 x 48 best, 8 x 24 and 16 x 12 slower;
            # layer4 at B = 32: 4 x 96), >= 16 in a tail beside whole workgroups (layer2: 6 x 16)
            min_st = 48 if gx == tail_x else 16
            fit = [s for s in cls.WINO2_PIECES if units % s == 0 and tail_x * ny * s <= 256 and 3 * (a.Cs // 4) // s >= min_st]
            s = forced or (max(fit) if fit else 2)
            return (gx - tail_x, s)
        if variant == 30:
            # Winograd: few-tile launches cut the K loop (9 x Cs / 8 stages) into as many even pieces as fill ONE residency round of
            # the 512 slots (2 workgroups / CU), each piece keeping >= 48 stages.  Measured at layer4, B = 32 (112 workgroups, 576
            # stages; scripts/r3/ab_split.sh): 4 pieces 167 / 171 TFLOP/s forward / data gradient, 8 pieces 157 / 161, 6 152 / 155,
            # 10 (two rounds and a fifth) 143 / 146, 3 129 / 131
            if _env("SLIC_WINO_SPLIT", "1") == "0":
                return None
            Wd = a.Ws
            gx, ny = ((a.M // Wd) * ((Wd + 3) // 4) + 63) // 64, a.N // 64
            wgs = gx * ny
            ns = 9 * (a.Cs // 8)
            forced = _env("SLIC_WINO_SPLIT", "1")
            if wgs >= int(_env("SLIC_WINO_MIN_WGS", cls.WINO_MIN_WGS)):
                # More than a round: a partly filled LAST dispatch round costs most of a round's time (layer2 at B = 32: 1568
                # workgroups on 512 slots = 3.06 rounds, 218 / 224 TFLOP/s against 236 / 237 at B = 31's 2.97 rounds).  The blocks
                # of the full rounds run whole; the few behind them cut their K loop so that they fill the slots once more.
                # Widths that are multiples of 4 (four GEMM rows per tile: the tail is a whole number of 128-row blocks).
                # Pieces of >= 36 stages (layer2: 4 x 36, 231 / 234; 6 x 24 and 3 x 48 gained less); launches of many rounds are
                # left alone (layer1: 12.25 rounds — cutting its 72-stage K loop cost the data gradient 3 %).
                rem = wgs % 512
                if forced == "0" or _env("SLIC_WINO_TAIL", "1") == "0" or Wd % 4 or rem == 0 or rem > 256 or wgs > 6 * 512:
                    return None
                tail_x = -(-rem // ny)
                s = min(512 // (tail_x * ny), ns // int(_env("SLIC_WINO_TAIL_MIN", "36")))
                return (gx - tail_x, s) if s >= 2 else None
            s = int(forced) if forced not in ("0", "1") else min(512 // wgs, ns // 48)
            return (0, s) if s >= 2 else None
        if variant not in (20, 22) or _env("SLIC_CONV_TAIL", "1") == "0":
            return None
        slots = int(_env("SLIC_CONV_TAIL_SLOTS", "0")) or cls.SLOTS[variant]
        BM = 128 if variant == 22 else 64
        nrb, ny = (a.M + BM - 1) // BM, (a.N + 63) // 64
        tiles, nk = nrb * ny, a.nchunks // 8
        full = (tiles // slots) * slots
        nfull_rb = full // ny
        tail_tiles = (nrb - nfull_rb) * ny
        if tail_tiles == 0 or tail_tiles * 4 > slots * 3:
            return None
        s = min(slots // tail_tiles, nk // cls.SPLIT_MIN_KTILES)
        return (nfull_rb, s) if s >= 2 else None

    def _launch(self, a, variant):
        variant = self._pick(a, variant)
        plan = self._plan_split(a, variant)
        if plan is not None:
            lib = _lib.load()
            nfull_rb, splits = plan
            ws = _lib.workspace(lib.slic_conv_gemm_tailsplit_workspace_bytes(ctypes.byref(a), variant, nfull_rb, splits),
                                self.device, "splitk")
            go = lambda: call("slic_conv_gemm_tailsplit", ctypes.byref(a), variant, nfull_rb, splits, ptr(ws), stream())
        else:
            go = lambda: call("slic_conv_gemm", ctypes.byref(a), variant, stream())
        if self.prof is None:
            go()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()                      # torch's current stream == the stream the kernel is launched on
        go()
        e1.record()
        self.prof.append((e0, e1, getattr(self, "_prof_tag", "fwd")))

    def dgrad(self, dz, wd, B, addend=None, out=None, variant=0, mask=None, bwd=None, skip_empty=False, addend_classes=None):
        """dz: [B, To, Ho, Wo, N] -> dx: [B, T, H, W, Cs] = conv_transpose(dz) (+ addend; addend may be `out`
        itself: every element is read and written by the same lane).
        mask (same shape as dx): dx = where(mask > 0, dx, 0) — the ReLU backward of the layer below, fused.
        bwd = (z, mean, invstd) of that layer's BatchNorm: also returns the per-workgroup partial sums
        (sum dx, sum dx * xhat) as a [R, 2, Cs] slab for slic_bn_bwd_fused -> returns (dx, partial).
        skip_empty: parity classes without a tap (a 1x1x1 stride-2 convolution reaches one position in eight) are NOT launched — their
        positions of dx stay unwritten, for a consumer that reads dx through addend_classes.
        addend_classes: the parity classes on which `addend` is defined (tap_classes() of the plan that wrote it); the other classes'
        launches take no addend."""
        lib = _lib.load()
        chunks = self._chunks(B)
        if chunks is not None:
            T, H, W = self.in_dims
            dx = out if out is not None else torch.empty((B, T, H, W, self.Cs), dtype=torch.float32, device=dz.device)
            parts = []
            for b0, b1 in chunks:
                r = self.dgrad(dz[b0:b1], wd, b1 - b0, None if addend is None else addend[b0:b1], dx[b0:b1], variant,
                               None if mask is None else mask[b0:b1], None if bwd is None else (bwd[0][b0:b1], bwd[1], bwd[2]),
                               skip_empty, addend_classes)
                if bwd is not None:
                    parts.append(r[1])
            return dx if bwd is None else (dx, torch.cat(parts))       # (sum dx, sum dx * xhat) partial rows: plain sums, any number of rows
        self._prof_tag = "dgrad"
        if self.wino:
            assert variant in (0, 30, 31), "a Winograd plan runs variant 30 / 31 only (build the plan with wino=False)"
            variant = 31 if self.wino2 else 30
        T, H, W = self.in_dims
        To, Ho, Wo = self.out_dims
        dx = out if out is not None else torch.empty((B, T, H, W, self.Cs), dtype=torch.float32, device=dz.device)
        launches = []
        # longest K loop first: the classes of a stride-2 layer have 1 ... 8 taps, one grid holds them all (blockIdx.z, dispatched
        # in order), and a launch that ends with its longest workgroups ends on a nearly empty chip
        for dc in sorted(self.dgrad_classes, key=lambda d: -d["nchunks"]):
            if skip_empty and dc["nchunks"] == 0:
                assert addend is None and mask is None and bwd is None
                continue
            a = SlicConvArgs()
            a.src = dz.data_ptr()
            a.src_bytes = _lib.u32_bytes(dz, 'dgrad source')
            a.wgt = wd.data_ptr()
            a.wgt_bytes = _lib.u32_bytes(wd, 'packed weights')
            a.dst = dx.data_ptr()
            a.tab = dc["tab"].data_ptr()
            a.tap_tab = dc["tap"].data_ptr() if dc["tap"] is not None else None
            a.addend = addend.data_ptr() if (addend is not None and (addend_classes is None or dc["cls"] in addend_classes)) else None
            a.mask_src = mask.data_ptr() if mask is not None else None
            ga, gb, gc = dc["grid"]
            a.M = B * ga * gb * gc
            a.N = self.Cs
            a.nchunks = dc["nchunks"]
            a.Cs, a.Ts, a.Hs, a.Ws = self.N, To, Ho, Wo
            a.Ga, a.Gb, a.Gc = ga, gb, gc
            a.sa = a.sb = a.sc = 1
            a.ldw, a.ldo = self.Kd, self.Cs
            strided = self.stride != (1, 1, 1)
            a.dst_strided = int(strided)
            a.Da, a.Db, a.Dc = T, H, W
            a.da, a.db, a.dc = self.stride
            a.ea, a.eb, a.ec = dc["cls"]
            launches.append(a)
        part = None
        if bwd is not None:
            z, mean, invstd = bwd
            assert z.shape == dx.shape and z.is_contiguous()
            def slab_rows(a):
                tm = lib.slic_conv_tile_m(ctypes.byref(a), self._pick(a, variant))
                if variant == 30 and self._plan_split(a, 30) is not None:
                    tm = 128                 # the split-K finish pass works on blocks of 128 real GEMM rows
                return (a.M + tm - 1) // tm
            rows = [slab_rows(a) for a in launches]
            part = torch.empty(sum(rows), 2, self.Cs, dtype=torch.float32, device=dz.device)
            r0 = 0
            for a, r in zip(launches, rows):
                a.bwd_z, a.bwd_mean, a.bwd_invstd = z.data_ptr(), mean.data_ptr(), invstd.data_ptr()
                a.bwd_partial = part.data_ptr() + r0 * 2 * self.Cs * 4
                r0 += r
        picks = [self._pick(a, variant) for a in launches]
        if len(launches) > 1 and len(set(picks)) == 1 and picks[0] in (20, 22):
            # the parity classes of a stride-2 layer as ONE launch: their K loops (1-8 taps) are too short to fill the chip
            # one class at a time
            arr = (SlicConvArgs * len(launches))(*launches)
            e0 = e1 = None
            if self.prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            call("slic_conv_gemm_multi", arr, len(launches), picks[0], stream())
            if self.prof is not None:
                e1.record()
                self.prof.append((e0, e1, "dgrad"))
        else:
            for a in launches:
                self._launch(a, variant)
        return dx if bwd is None else (dx, part)

    def tap_classes(self):
        """the parity classes of the data gradient that have at least one tap (the positions dgrad(skip_empty=True) writes)"""
        return {dc["cls"] for dc in self.dgrad_classes if dc["nchunks"] > 0}

    def _row_table(self, a, B):
        """per-row {source byte offset, in-bounds mask} records of the forward geometry at batch B (8 bytes / row),
        built once on the device and kept with the plan"""
        t = self._row_tabs.get(B)
        if t is None:
            t = torch.empty(a.M, 2, dtype=torch.int32, device=self.device)
            call("slic_conv_row_table", ctypes.byref(a), ptr(t), stream())
            self._row_tabs[B] = t
        return t

    @staticmethod
    def _wino_wgrad_slices(blocks, mt):
        """tile slices of the transposed-Winograd weight gradient: `blocks` = 9 (kt, kh) x 64 x 64 blocks, `mt` W-tiles.
        Up to 512 blocks: one or two residency rounds of the 512 slots (2 workgroups / CU), at least 64 tiles per slice.  More blocks than
        slots (layer4: 576 = one round and an eighth): each slice costs a 6-point slab of 2 x the weight's size to write and re-read,
        so only 1-3 slices are weighed — how full the rounds are x the main loop's share of a workgroup's time, less the slab traffic
        (measured at layer4, B = 32: 1 slice 128 TFLOP/s, 2 143, 3 144, 4 131, 7 106; scripts/r3/ab_split.sh)."""
        forced = _env("SLIC_WINO_WGRAD_WGS")
        if forced is not None:
            return max(1, min(int(forced) // blocks, mt // 64))
        if blocks <= 512:
            # ONE residency round when the slices fill it (layer1: 9 blocks x 56 slices = 504 workgroups, layer2: 36 x 14 — half the
            # slab traffic of two rounds, +1-2 %), else two (layer3: 144 x 7 = 1008; one round would be 3 slices = 432, 170 against 191)
            one = 512 // blocks
            return max(1, min(one if one * blocks >= 486 else 1024 // blocks, mt // 64))
        best, best_score = 1, -1.0
        for s in (1, 2, 3):
            if mt // s < 64:
                break
            wgs, stages = blocks * s, mt / s / 8.0
            fill = wgs / (-(-wgs // 512) * 512.0)
            score = fill * stages / (stages + 8.0) - 0.06 * s
            if score > best_score:
                best, best_score = s, score
        return best

    def wgrad(self, x, dz, B, dW, splits=None):
        """dW (reference layout [N, C, kt, kh, kw], written in place) = gather(x)^T dz"""
        lib = _lib.load()
        chunks = self._chunks(B)
        if chunks is not None:
            tmp = torch.empty_like(dW)
            for i, (b0, b1) in enumerate(chunks):                      # chunk order: a fixed summation order
                self.wgrad(x[b0:b1], dz[b0:b1], b1 - b0, dW if i == 0 else tmp, splits)
                if i:
                    dW.add_(tmp)
            return dW
        a = self._fwd_args(x, B)
        if self.wino2_wgrad:
            # transposed F(4, 3) x F(2, 3): one workgroup of 512 threads per kt, 64 x 64 block and slice of the 2 x 4
            # tiles — ONE residency round of the 256 slots (one workgroup per CU), at least 64 tiles per slice
            blocks = 3 * (self.C // 64) * (self.N // 64)
            H2, W2 = self.in_dims[1], self.in_dims[2]
            mt = (a.M // (H2 * W2)) * ((H2 + 1) // 2) * ((W2 + 3) // 4)
            if splits is None:
                forced = _env("SLIC_WINO2_WGRAD_WGS")
                splits = max(1, min((int(forced) if forced else 256) // blocks, mt // 64))
            tab = self._wino2_tabs.get(B)
            if tab is None:
                tab = torch.empty(mt, 2, dtype=torch.int32, device=self.device)
                call("slic_conv_wino2_tile_table", ctypes.byref(a), ptr(tab), stream())
                self._wino2_tabs[B] = tab
            ws = _lib.workspace(lib.slic_conv_wgrad_wino2_workspace_bytes(ctypes.byref(a), splits), x.device, "wgrad")
            call("slic_conv_wgrad_wino2", ctypes.byref(a), ptr(dz), splits, ptr(tab), ptr(dW), ptr(ws), stream())
            return dW
        if self.wino_wgrad:
            # transposed F(4, 3): one workgroup per (kt, kh), 64 x 64 block and slice of the W-tiles; ~2 residency rounds of the
            # 512 slots (2 workgroups / CU), at least 64 tiles per slice
            blocks = 9 * (self.C // 64) * (self.N // 64)
            mt = (a.M // self.in_dims[2]) * ((self.in_dims[2] + 3) // 4)
            if splits is None:
                splits = self._wino_wgrad_slices(blocks, mt)
            tab = self._wino_tabs.get(B)
            if tab is None:
                tab = torch.empty(mt, 2, dtype=torch.int32, device=self.device)
                call("slic_conv_wino_tile_table", ctypes.byref(a), ptr(tab), stream())
                self._wino_tabs[B] = tab
            ws = _lib.workspace(lib.slic_conv_wgrad_wino_workspace_bytes(ctypes.byref(a), splits), x.device, "wgrad")
            call("slic_conv_wgrad_wino", ctypes.byref(a), ptr(dz), splits, ptr(tab), ptr(dW), ptr(ws), stream())
            return dW
        if splits is None:
            # measured (scripts/bench_conv.py, WGONLY=1 sweep): 128 x 64 output tiles, ~3000 workgroups, but at
            # least 1024 positions per slice so the slabs of the small-M layers stay small
            blocks = ((self.nchunks_fwd + 31) // 32) * ((self.N + 63) // 64)
            splits = max(1, min((3072 + blocks - 1) // blocks, (a.M + 1023) // 1024))
        a.row_tab = self._row_table(a, B).data_ptr()
        ws = _lib.workspace(lib.slic_conv_wgrad_workspace_bytes(ctypes.byref(a), splits), x.device, "wgrad")
        call("slic_conv_wgrad", ctypes.byref(a), ptr(dz), self.N, splits, self.C, self.ntaps, ptr(dW), ptr(ws), stream())
        return dW
