"""TEST-ONLY kernel provider for FINCH(kernels=...): the HipFinchKernels interface in NumPy on host arrays, so the host
logic of clustering/finch.py (link list, weighted early-exit cut, hierarchy bookkeeping, req_clust refinement) runs
against the reference goldens on a GPU-less machine.  Never shipped, never imported by the package."""
import numpy as np


class NumpyFinchKernels:
    def resident(self, mat):
        return np.ascontiguousarray(np.asarray(mat, dtype=np.float32))

    @staticmethod
    def _unit(rows):
        r = rows.astype(np.float64)
        return r / np.maximum(np.linalg.norm(r, axis=1, keepdims=True), 1e-300)

    def first_neighbours(self, rows):
        n = rows.shape[0]
        if n == 1:
            return np.zeros(1, np.int64)
        u = self._unit(rows)
        out = np.empty(n, np.int64)
        for s in range(0, n, 2048):
            sim = u[s:s + 2048] @ u.T
            sim[np.arange(sim.shape[0]), s + np.arange(sim.shape[0])] = -np.inf
            out[s:s + 2048] = sim.argmax(axis=1)
        return out

    def pair_cosine_distance(self, rows, a, b):
        u = self._unit(rows)
        return (1.0 - np.einsum("ij,ij->i", u[a], u[b])).astype(np.float32)

    def cluster_means(self, rows, assign, K):
        sums = np.zeros((K, rows.shape[1]), np.float32)
        np.add.at(sums, assign, rows)                          # ascending row order, fp32 adds (km_accumulate's order)
        counts = np.bincount(assign, minlength=K).astype(np.float32)
        return sums / counts[:, None]
