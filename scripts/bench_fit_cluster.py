"""GPU box: end-to-end seconds of the reference-shaped call fit_cluster(emb, 'kmeans', k=500) = KMeans(n_clusters=500, n_init=10)
(k-means++ x 10, max_iter 300, tol 1e-4) at 100k x 512; and FINCH at the same size"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_similarity_search_amd.clustering import fit_cluster
N, D, K = 100000, 512, 500
rng = np.random.default_rng(1)
cent = rng.standard_normal((K, D)); cent /= np.linalg.norm(cent, axis=1, keepdims=True)
z = rng.integers(0, K, N)
X = (cent[z] + 0.35 * rng.standard_normal((N, D)) / np.sqrt(D)).astype(np.float32)
Xd = torch.from_numpy(X).cuda()
np.random.seed(1)
for n_init in (1, 10):
    torch.cuda.synchronize(); t0 = time.time()
    lab = fit_cluster(Xd, 'kmeans', k=K, l2normalize=True, n_init=n_init)
    torch.cuda.synchronize(); dt = time.time() - t0
    km = fit_cluster.last_model
    from sklearn.metrics import normalized_mutual_info_score as nmi
    print(f"n_init={n_init}: {dt:.2f} s  inertia {km.inertia_:.3f} n_iter(best) {km.n_iter_} NMI vs truth {nmi(z, lab):.4f}", flush=True)
t0 = time.time(); lab = fit_cluster(Xd, 'finch', finch_partition=1); print(f"finch: {time.time()-t0:.2f} s, {len(set(lab.tolist()))} clusters")
