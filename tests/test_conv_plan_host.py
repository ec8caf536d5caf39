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
