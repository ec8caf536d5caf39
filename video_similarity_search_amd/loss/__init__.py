from .triplet_loss import OnlineTripletLoss, MemTripletLoss, pdist, pdist_v2  # noqa: F401
from .NCE_loss import NCEAverage, NCESoftmaxLoss, NCECriterion, AliasMethod  # noqa: F401
