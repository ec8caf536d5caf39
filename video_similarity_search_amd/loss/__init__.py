from .triplet_loss import OnlineTripletLoss, pdist  # noqa: F401
from .NCE_loss import NCEAverage, NCESoftmaxLoss, AliasMethod  # noqa: F401
