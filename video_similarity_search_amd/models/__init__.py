from .resnet import generate_model, ResNet, BasicBlock  # noqa: F401
from .triplet_net import Tripletnet  # noqa: F401
