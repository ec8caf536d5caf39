"""host-side launch planning of the gather-GEMM (no GPU needed)"""
import os

from video_similarity_search_amd.models.conv_plan import ConvPlan
from video_similarity_search_amd._lib import SlicConvArgs


def test_plan_split_rules():
    """host-side rule (no GPU): R3D-18 at B = 32 — layer1 and layer3 split their last partial round, layer4 is all tail,
    layer2's remainder (0.9 of a round) is left alone"""
    def args(M, N, K):
        a = SlicConvArgs()
        a.M, a.N, a.nchunks = M, N, K // 4
        return a
    os.environ.pop("SLIC_CONV_TAIL", None)
    os.environ.pop("SLIC_CONV_TAIL_SLOTS", None)
    assert ConvPlan._plan_split(args(1605632, 64, 1728), 22) == (12288, 3)
    assert ConvPlan._plan_split(args(200704, 128, 3456), 20) is None
    assert ConvPlan._plan_split(args(25088, 256, 6912), 20) == (320, 4)
    assert ConvPlan._plan_split(args(3136, 512, 13824), 20) == (0, 3)
    assert ConvPlan._plan_split(args(3136, 512, 13824), 0) is None


def test_winograd_split_rules(monkeypatch):
    """host-side rules of the Winograd launches at R3D-18's shapes, B = 32 (models/conv_plan.py: _plan_split variant 30,
    _wino_wgrad_slices): layer1 (12.25 dispatch rounds) is left alone, layer2 (3.06 rounds) runs three whole rounds and cuts the 16
    tile blocks behind them 4 ways, layer3 (448 workgroups, under a round, ragged width) is left alone, layer4 (112 workgroups) cuts
    its K loop into 4 even pieces; weight-gradient slices fill one round where they can (layers 1, 2), two otherwise (layer3), and
    layer4's 576 blocks take 2 slices"""
    for k in ("SLIC_WINO_SPLIT", "SLIC_WINO_TAIL", "SLIC_WINO_TAIL_MIN", "SLIC_WINO_MIN_WGS", "SLIC_WINO_WGRAD_WGS"):
        monkeypatch.delenv(k, raising=False)

    def args(M, W, N, C):
        a = SlicConvArgs()
        a.M, a.Ws, a.N, a.Cs = M, W, N, C
        return a
    B = 32
    assert ConvPlan._plan_split(args(B * 16 * 56 * 56, 56, 64, 64), 30) is None
    assert ConvPlan._plan_split(args(B * 8 * 28 * 28, 28, 128, 128), 30) == (768, 4)
    assert ConvPlan._plan_split(args(B * 4 * 14 * 14, 14, 256, 256), 30) is None
    assert ConvPlan._plan_split(args(B * 2 * 7 * 7, 7, 512, 512), 30) == (0, 4)
    # a batch whose layer2 launch is two rounds and five workgroups: the three tile blocks behind the two rounds are cut
    nfull, s = ConvPlan._plan_split(args(21 * 8 * 28 * 28, 28, 128, 128), 30)
    assert nfull == 512 and 2 <= s <= 4
    monkeypatch.setenv("SLIC_WINO_TAIL", "0")
    assert ConvPlan._plan_split(args(B * 8 * 28 * 28, 28, 128, 128), 30) is None
    monkeypatch.delenv("SLIC_WINO_TAIL")
    monkeypatch.setenv("SLIC_WINO_SPLIT", "0")
    assert ConvPlan._plan_split(args(B * 2 * 7 * 7, 7, 512, 512), 30) is None
    monkeypatch.delenv("SLIC_WINO_SPLIT")
    assert [ConvPlan._wino_wgrad_slices(b, mt) for b, mt in ((9, 401408), (36, 50176), (144, 7168), (576, 896))] == [56, 14, 7, 2]
    assert ConvPlan._wino_wgrad_slices(9, 200) == 3            # never fewer than 64 tiles per slice
    monkeypatch.setenv("SLIC_WINO_WGRAD_WGS", "1024")
    assert ConvPlan._wino_wgrad_slices(9, 401408) == 113


def test_winograd_2d_plan_rules(monkeypatch):
    """host-side rules of the two-dimensional Winograd plans at R3D-18's shapes (models/conv_plan.py): variant 31 where the blocks of 64
    tiles are uniform and the launch has at least 64 workgroups; its K split by kt for a partly filled last dispatch round (layer2 at
    B = 32: 784 workgroups = three rounds + 16) and for launches of less than half a round (layer4: 64); the transposed two-dimensional
    weight gradient wherever its 3 x C/64 x N/64 workgroups per slice fit the 256 slots (all four layers; layer4: 192, one slice); channel counts that are not 64 x a power of two stay off the Winograd kernels"""
    for k in ("SLIC_WINO", "SLIC_WINO2", "SLIC_WINO2_MIN_WGS", "SLIC_WINO2_SPLIT", "SLIC_WINO2_WGRAD", "SLIC_WINO_WGRAD", "SLIC_WINO2_PERSIST",
              "SLIC_WINO2_PERSIST_GRID", "SLIC_WINO2_HALFTAIL"):
        monkeypatch.delenv(k, raising=False)
    k3, s1, p1 = (3, 3, 3), (1, 1, 1), (1, 1, 1)
    shapes = [(64, (16, 56, 56)), (128, (8, 28, 28)), (256, (4, 14, 14)), (512, (2, 7, 7))]
    plans = {C: ConvPlan(C, C, k3, s1, p1, dims, "cpu", batch=32) for C, dims in shapes}
    assert [plans[C].wino2 for C, _ in shapes] == [True, True, True, True]
    assert [plans[C].wino2_wgrad for C, _ in shapes] == [True, True, True, True]
    monkeypatch.setenv("SLIC_WINO2_WGRAD_MAXBLOCKS", "128")
    assert [ConvPlan(C, C, k3, s1, p1, dims, "cpu", batch=32).wino2_wgrad for C, dims in shapes] == [True, True, True, False]
    monkeypatch.delenv("SLIC_WINO2_WGRAD_MAXBLOCKS")
    assert all(plans[C].wino and plans[C].wino_wgrad for C, _ in shapes)

    def args(B, C, dims):
        a = SlicConvArgs()
        T, H, W = dims
        a.M, a.Ts, a.Hs, a.Ws, a.N, a.Cs = B * T * H * W, T, H, W, C, C
        return a
    assert ConvPlan._plan_split(args(32, 64, (16, 56, 56)), 31) is None          # 3136 workgroups: 12.25 rounds, remainder 64 ... of 256
    # 392 tile blocks x 2 = three rounds + 16: round 6 leaves the launch whole — the persistent kernel ends it in column-half items — and
    # with the persistent kernel (or its half items) off the last 8 blocks are cut into 6 pieces as in round 5
    assert ConvPlan._plan_split(args(32, 128, (8, 28, 28)), 31) is None
    for var in ("SLIC_WINO2_HALFTAIL", "SLIC_WINO2_PERSIST"):
        monkeypatch.setenv(var, "0")
        assert ConvPlan._plan_split(args(32, 128, (8, 28, 28)), 31) == (384, 6)
        monkeypatch.delenv(var)
    assert ConvPlan._plan_split(args(32, 256, (4, 14, 14)), 31) is None          # 224 workgroups: most of a round
    assert ConvPlan._plan_split(args(32, 512, (2, 7, 7)), 31) == (0, 4)          # 64 workgroups: all of them cut, 4 pieces = one dispatch round
    # small batches: layer3 at B = 8 has 56 workgroups -> still two-dimensional (4 K pieces fill the slots); layer4's 16 -> the one-dimensional
    # forward / data gradient, but the two-dimensional weight gradient (it needs 128 tiles, not a full dispatch round)
    p3 = ConvPlan(256, 256, k3, s1, p1, (4, 14, 14), "cpu", batch=8)
    assert p3.wino2 and p3.wino2_wgrad and ConvPlan._plan_split(args(8, 256, (4, 14, 14)), 31) == (0, 4)
    p4 = ConvPlan(512, 512, k3, s1, p1, (2, 7, 7), "cpu", batch=8)
    assert p4.wino and not p4.wino2 and p4.wino2_wgrad
    assert not ConvPlan(512, 512, k3, s1, p1, (2, 7, 7), "cpu", batch=2).wino2_wgrad       # 32 tiles
    # no batch hint (a plan built by hand): never variant 31 by default
    assert not ConvPlan(64, 64, k3, s1, p1, (16, 56, 56), "cpu").wino2
    # widths 192 / 384 (RESNET.WIDEN_FACTOR 1.5 / 3): the Winograd forward / data gradient address their K loop with shifts
    for C in (192, 384):
        pl = ConvPlan(C, C, k3, s1, p1, (4, 14, 14), "cpu", batch=32)
        assert not pl.wino and not pl.wino2 and pl.wino_wgrad
    # non-uniform blocks (odd height whose tiles per frame do not divide 64): not eligible
    assert not ConvPlan(64, 64, k3, s1, p1, (4, 9, 20), "cpu", batch=64).wino2
