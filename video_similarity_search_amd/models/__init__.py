from .resnet import generate_model, ResNet, BasicBlock  # noqa: F401
from .triplet_net import Tripletnet  # noqa: F401
from .r3d import R3DNet, r3d_model  # noqa: F401
