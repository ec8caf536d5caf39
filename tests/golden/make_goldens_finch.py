"""
Generates tests/golden/finch.npz in the BUILD container by importing the reference's clustering/finch.py
(importable: numpy/scipy/sklearn only; pyflann absent, so N <= 70 000).
    python tests/golden/make_goldens_finch.py
"""
import os
import sys
sys.dont_write_bytecode = True
import warnings
warnings.simplefilter("ignore")
import numpy as np
sys.path.insert(0, "/root/reference/clustering")
from finch import FINCH     # noqa: E402  (the reference)

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(31)
N, D, C = 3000, 64, 40
cent = rng.standard_normal((C, D))
cent /= np.linalg.norm(cent, axis=1, keepdims=True)
z = rng.integers(0, C, N)
X = (cent[z] + 0.45 * rng.standard_normal((N, D)) / np.sqrt(D)).astype(np.float32)
c, num_clust, req_c = FINCH(X, distance='cosine', verbose=False)
c2, nc2, req = FINCH(X, req_clust=25, distance='cosine', verbose=False)
np.savez_compressed(os.path.join(HERE, "finch.npz"), X=X, z=z.astype(np.int32), c=c.astype(np.int32),
                    num_clust=np.array(num_clust), req_c=req.astype(np.int32), req_clust=25)
print("partitions", num_clust, "req", len(np.unique(req)))
