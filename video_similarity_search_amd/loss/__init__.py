from .triplet_loss import OnlineTripletLoss, pdist  # noqa: F401
