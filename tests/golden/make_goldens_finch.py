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
out = dict(X=X, z=z.astype(np.int32), c=c.astype(np.int32), num_clust=np.array(num_clust), req_c=req.astype(np.int32), req_clust=25)
print("partitions", num_clust, "req", len(np.unique(req)))

# a deeper hierarchy (many small, unevenly spread groups -> several levels) and a req_clust below the coarsest partition's neighbour
rng2 = np.random.default_rng(32)
N2, D2, C2 = 2500, 32, 300
cent2 = rng2.standard_normal((C2, D2)) + 2.0 * rng2.standard_normal((1, D2))
z2 = rng2.integers(0, C2, N2)
X2 = (cent2[z2] + 0.25 * rng2.standard_normal((N2, D2))).astype(np.float32)
cD, ncD, _ = FINCH(X2, distance='cosine', verbose=False)
_, _, reqD = FINCH(X2, req_clust=7, distance='cosine', verbose=False)
cN, ncN, _ = FINCH(X2, distance='cosine', ensure_early_exit=False, verbose=False)
out.update(deep_X=X2, deep_c=cD.astype(np.int32), deep_num_clust=np.array(ncD), deep_req_c=reqD.astype(np.int32), deep_req_clust=7,
           deep_noexit_c=cN.astype(np.int32), deep_noexit_num_clust=np.array(ncN))
print("deep", ncD, "no early exit", ncN, "req", len(np.unique(reqD)))

# degenerate: three points whose first-neighbour graph is one component already at level 0
X3 = (np.ones((3, 8)) + 0.05 * np.random.default_rng(33).standard_normal((3, 8))).astype(np.float32)
c3, nc3, _ = FINCH(X3, distance='cosine', verbose=False)
out.update(one_X=X3, one_c=c3.astype(np.int32), one_num_clust=np.array(nc3))
print("one", nc3, c3.shape)

# datasets on which the early-exit cut DROPS links, and on which the reference's adjacency weight matters (a mutual
# first-neighbour pair stands at 2 d in the bound and in the cut, clustering/finch.py:38-40,44-45,144): with the plain
# distance in its place the partitions differ.  Several seeds, small.
for t, seed in enumerate((4, 7, 32)):
    r = np.random.default_rng(seed)
    Nc, Dc, Cc = 600, 16, 30
    cc = r.standard_normal((Cc, Dc))
    zc = r.integers(0, Cc, Nc)
    Xc = (cc[zc] + 0.5 * r.standard_normal((Nc, Dc))).astype(np.float32)
    ce, nce, _ = FINCH(Xc, distance='cosine', verbose=False)
    cn_, ncn_, _ = FINCH(Xc, distance='cosine', ensure_early_exit=False, verbose=False)
    assert list(nce) != list(ncn_), "the cut must drop a link on this dataset"
    _, _, rq = FINCH(Xc, req_clust=10, distance='cosine', verbose=False)
    out.update({f"cut{t}_X": Xc, f"cut{t}_c": ce.astype(np.int32), f"cut{t}_num_clust": np.array(nce),
                f"cut{t}_noexit_num_clust": np.array(ncn_), f"cut{t}_req_c": rq.astype(np.int32)})
    print("cut", seed, nce, "no early exit", ncn_)
np.savez_compressed(os.path.join(HERE, "finch.npz"), **out)
