"""Host logic of ConvPlan's size gates (no GPU: the plan's tables are built on the CPU).

The two-dimensional Winograd kernels address their source through a 32-bit raw buffer whose out-of-range offset 0xFFFFFF00 stands for
padding pixels (csrc/conv_wino2.hip: wino2_check) and whose weight gradient holds 24-bit positions (slic_conv_wgrad_wino2): a batch
beyond ONE launch's range must be cut into chunks of whole clips, every chunk inside both limits, and a plan never claims variant 31
for a launch that does not fit.  These are the cuDNN calls behind nn.Conv3d of /root/reference/models/resnet.py:11-17, which have no
such limit — hence the chunking instead of an error.
"""
import numpy as np
import pytest
import torch

from video_similarity_search_amd import _lib
from video_similarity_search_amd.models.conv_plan import ConvPlan

OOB = 0xFFFFFF00


def _plan(C, N, dims, batch, **kw):
    return ConvPlan(C, N, (3, 3, 3), (1, 1, 1), (1, 1, 1), dims, torch.device("cpu"), batch=batch, **kw)


def _fits(plan, clips):
    T, H, W = plan.in_dims
    cmax = max(plan.Cs, plan.N)
    pos = clips * T * H * W
    return pos * cmax * 4 + 2 * H * W * cmax * 4 + 16 <= OOB and pos < (1 << 24)


@pytest.mark.parametrize("dims,C", [((16, 56, 56), 64), ((16, 64, 64), 64), ((8, 28, 28), 128), ((16, 80, 80), 64)])
@pytest.mark.parametrize("B", [8, 32, 165, 312, 400, 1000])
def test_chunks_stay_inside_one_launch_range(dims, C, B):
    p = _plan(C, C, dims, B)
    ch = p._chunks(B)
    if ch is None:
        assert _fits(p, B), "a batch run as one launch must fit the 32-bit range and the 24-bit positions"
        return
    # contiguous cover of [0, B), whole slab rows (multiples of 8 clips) but for the last chunk
    assert ch[0][0] == 0 and ch[-1][1] == B and all(a[1] == b[0] for a, b in zip(ch, ch[1:]))
    assert all((e - s) % 8 == 0 for s, e in ch[:-1]) and all(e > s for s, e in ch)
    assert all(_fits(p, e - s) for s, e in ch)
    # the plan's own gates saw the clips of one launch, not the whole batch
    assert p.wino2_size_ok


def test_layer1_limits_at_112_and_128():
    # layer1 of R3D-18 (64 -> 64, 16 x 56 x 56 after the stem at 112 x 112): one launch holds the 2^24 - 1 positions of 334 clips at most
    p = _plan(64, 64, (16, 56, 56), 32)
    assert p._chunks(32) is None and p._chunks(328) is None and p.wino2 and p.wino2_wgrad and p.wino2_size_ok
    assert p._launch_batch(400) == 328 and _fits(p, 328) and not _fits(p, 336)
    # 128 x 128 clips: 16 x 64 x 64 at layer1; configs[3]'s 312-clip global batch on one GPU runs as 248 + 64
    p = _plan(64, 64, (16, 64, 64), 312)
    assert p._chunks(312) == [(0, 248), (248, 312)]


def test_out_of_range_offset_is_never_in_range():
    # the padding offset must lie beyond source + one frame on either side for every launch the plan admits
    for dims, C in [((16, 56, 56), 64), ((16, 64, 64), 64), ((8, 32, 32), 128), ((4, 16, 16), 256)]:
        p = _plan(C, C, dims, 10 ** 6)
        Bc = p._launch_batch(10 ** 6)
        T, H, W = dims
        num_records = Bc * T * H * W * C * 4 + 2 * H * W * C * 4
        assert num_records + 16 <= OOB and num_records < (1 << 32)
        assert Bc * T * H * W < (1 << 24)


def test_forced_limits_chunk_or_raise(monkeypatch):
    monkeypatch.setenv("SLIC_CONV_MAX_POSITIONS", str(16 * 16 * 16 * 16))          # 16 clips of 16^3 positions
    p = _plan(64, 64, (16, 16, 16), 40)
    assert p._chunks(40) == [(0, 16), (16, 32), (32, 40)]
    monkeypatch.setenv("SLIC_CONV_MAX_POSITIONS", str(16 * 16 * 16 * 7))           # fewer than eight clips per launch: nothing to run
    with pytest.raises(_lib.SlicError):
        _plan(64, 64, (16, 16, 16), 40)._chunks(40)
    monkeypatch.delenv("SLIC_CONV_MAX_POSITIONS")
    monkeypatch.setenv("SLIC_CONV_MAX_BYTES", str(24 * 16 * 16 * 16 * 64 * 4 + 2 * 16 * 16 * 64 * 4 + 256))
    assert _plan(64, 64, (16, 16, 16), 40)._launch_batch(40) == 24


def test_widths_outside_the_winograd_grid_fall_back():
    # 64 x a power of two channels only (the K loop addresses 8-channel stages with a shift): 192 / 384 stay on the direct kernels
    for C in (192, 384):
        p = _plan(C, C, (8, 28, 28), 32)
        assert not p.wino and not p.wino2 and not p.wino2_wgrad
    # a width whose padded form does not divide 128 (W = 20 -> 20; 128 % 20 != 0 but 20 % 4 == 0: eligible) vs W = 18 (pad 20: not)
    assert _plan(64, 64, (8, 20, 20), 32).wino
    assert not _plan(64, 64, (8, 18, 18), 32).wino
    # few workgroups: the two-dimensional forward is not claimed for a launch that cannot fill the chip
    assert not _plan(512, 512, (2, 7, 7), 2).wino2
