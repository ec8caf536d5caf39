"""
Generates tests/golden/retrieval.npz in the BUILD container: the reference's retrieval arithmetic is
sklearn.metrics.pairwise.cosine_distances + np.argsort (iic_retrieve_clips.py:295-296) and
cosine_distances + argpartition (evaluate.py:213,226-231); evaluate.py / iic_retrieve_clips.py cannot be imported
here (cv2 / torchvision), so the goldens come from those sklearn/numpy calls, transcribed 1:1.
    python tests/golden/make_goldens_retrieval.py
"""
import os
import numpy as np
from sklearn.metrics.pairwise import cosine_distances

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(5)
Nq, Ng, D, C = 300, 2000, 64, 17
# class-structured features so hit rates are informative
cent = rng.standard_normal((C, D))
y_train = rng.integers(0, C, Ng)
y_test = rng.integers(0, C, Nq)
X_train = (cent[y_train] + 2.0 * rng.standard_normal((Ng, D))).astype(np.float64)     # float64 like the .npy files (D7)
X_test = (cent[y_test] + 2.0 * rng.standard_normal((Nq, D))).astype(np.float64)
ks = [1, 5, 10, 20, 50]
distances = cosine_distances(X_test, X_train)
indices = np.argsort(distances)
topk_correct = {k: 0 for k in ks}
for k in ks:
    for ind, test_label in zip(indices[:, :k], y_test):
        if test_label in y_train[ind]:
            topk_correct[k] += 1
# evaluate.py flavour: float32, self-retrieval with the diagonal at +inf, top-20 accuracies
Xs = X_train[:500].astype(np.float32)
dm = cosine_distances(Xs)
np.fill_diagonal(dm, float('inf'))
idx = np.argpartition(dm, 20, axis=-1)
un = np.take_along_axis(dm, idx[:, :20], axis=-1)
top20 = np.take_along_axis(idx, np.argsort(un, axis=-1), axis=-1)
acc = []
ys = y_train[:500]
for k in [1, 5, 10, 20]:
    acc.append(np.mean([ys[i] in ys[top20[i, :k]] for i in range(500)]))
np.savez_compressed(os.path.join(HERE, "retrieval.npz"), X_train=X_train, y_train=y_train, X_test=X_test, y_test=y_test,
                    ks=np.array(ks), topk_correct=np.array([topk_correct[k] for k in ks]), top50=indices[:, :50].astype(np.int32),
                    d50=np.take_along_axis(distances, indices[:, :50], axis=1), self_top20=top20.astype(np.int32),
                    self_acc=np.array(acc), self_dm=dm.astype(np.float32))
print(topk_correct, acc)
