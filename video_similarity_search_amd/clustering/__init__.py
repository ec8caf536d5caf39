from .cluster_masks import fit_cluster, preprocess_features_kmeans  # noqa: F401
from .kmeans_hip import KMeans  # noqa: F401
