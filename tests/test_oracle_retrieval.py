"""CPU: the retrieval oracle vs goldens made with sklearn's cosine_distances + argsort (the reference's calls)."""
import os
import numpy as np
from oracle import retrieval as orr


def test_retrieval_oracle_matches_sklearn_golden(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "retrieval.npz")))
    hits, ind = orr.topk_retrieval(g["X_train"], g["y_train"], g["X_test"], g["y_test"], ks=list(g["ks"]))
    assert [hits[int(k)] for k in g["ks"]] == list(g["topk_correct"])
    assert (ind[:, :50] == g["top50"]).mean() > 0.999          # identical except exact ties
    acc, ind2 = orr.topk_acc_self(g["X_train"][:500].astype(np.float32), g["y_train"][:500])
    np.testing.assert_allclose(acc, g["self_acc"], atol=1e-12)
