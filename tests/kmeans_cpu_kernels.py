"""TEST-ONLY kernel provider for KMeans(kernels=...): the HipKernels interface implemented with the CPU oracle on CPU
tensors, so the multi-process control flow of the sharded k-means (collectives, ordered combine, relocation merge,
convergence bookkeeping) can run under gloo on a GPU-less host.  Never shipped, never imported by the package."""
import numpy as np
import torch

from oracle import kmeans as ok


def _np(t):
    return t.detach().numpy()


class OracleKernels:
    device_type = "cpu"

    def check(self):
        pass

    def to_device(self, X):
        if not torch.is_tensor(X):
            X = torch.as_tensor(np.ascontiguousarray(X, dtype=np.float32))
        return X.detach().to(dtype=torch.float32).contiguous()

    def col_stats(self, X):
        x = _np(X).astype(np.float64)
        N, D = x.shape
        s1, s2 = np.zeros(D), np.zeros(D)
        for s in range(0, N, 1024):                      # 1024-row segments, rows ascending (kmeans.hip col_stats_seg)
            seg = x[s:s + 1024]
            a, b = np.zeros(D), np.zeros(D)
            for row in seg:
                a += row
                b += row * row
            s1 += a
            s2 += b
        return torch.from_numpy(np.stack([s1, s2]))

    def sub_rowvec(self, X, v, out):
        out.copy_(X - v[None, :])

    def cnorm(self, C, cnorm):
        cnorm.copy_(torch.from_numpy(ok.row_sqnorm_chain(_np(C))))

    def assign(self, X, C, cnorm, labels, labels_old, n_changed):
        lab = torch.from_numpy(ok.assign(_np(X), _np(C)))
        if labels_old is not None:
            n_changed += int((lab != labels_old).sum())
        labels.copy_(lab)

    def accumulate(self, X, labels, K, sums, counts):
        s, c = ok.accumulate(_np(X), _np(labels), K, 1)
        sums.copy_(torch.from_numpy(s.reshape(-1)))
        counts.copy_(torch.from_numpy(c))

    def combine_shards(self, allpart, K, Dp, sums, counts):
        ap = _np(allpart)
        s = np.zeros(K * Dp, np.float32)
        c = np.zeros(K, np.float32)
        for r in range(ap.shape[0]):                     # rank order, fp32 adds (km_combine_shards)
            s = s + ap[r, :K * Dp]
            c = c + ap[r, K * Dp:]
        sums.copy_(torch.from_numpy(s))
        counts.copy_(torch.from_numpy(c))

    def lloyd_local(self, X, Xp, C_old, Cp_old, cnorm_old, labels, labels_old, payload):
        """slic_kmeans_lloyd_local: E-step + ordered M-step -> [K*D sums | K counts | n_changed & 0xFFFFF | n_changed >> 20]"""
        K, Dp = C_old.shape
        lab = torch.from_numpy(ok.assign(_np(X), _np(C_old)))
        nc = int((lab != labels_old).sum()) if labels_old is not None else 0
        labels.copy_(lab)
        s, c = ok.accumulate(_np(X), _np(labels), K, 1)
        flat = payload.view(-1)
        flat[:K * Dp] = torch.from_numpy(s.reshape(-1)).to(payload.dtype)
        flat[K * Dp:K * Dp + K] = torch.from_numpy(c).to(payload.dtype)
        flat[K * Dp + K] = float(nc & 0xFFFFF)
        flat[K * Dp + K + 1] = float(nc >> 20)

    def lloyd_global(self, parts, C_old, sums, counts, C_new, Cp_new, cnorm_new, shift, status, spherical=False):
        """slic_kmeans_lloyd_global: fp32 payloads added in part order / the reduced fp64 payload rounded to fp32, then finalize"""
        K, Dp = C_old.shape
        ap = _np(parts)
        if ap.dtype == np.float64:
            tot = np.zeros(ap.shape[1], np.float64)
            for r in range(ap.shape[0]):
                tot = tot + ap[r]
            comb = tot.astype(np.float32)
            nc = float(round(tot[K * Dp + K] + 1048576.0 * tot[K * Dp + K + 1]))      # a double, as km_status computes it (a poisoned payload: 2^60)
        else:
            comb = np.zeros(ap.shape[1], np.float32)
            for r in range(ap.shape[0]):                 # rank order, fp32 adds
                comb = comb + ap[r]
            nc = float(round(float(comb[K * Dp + K]) + 1048576.0 * float(comb[K * Dp + K + 1])))
        sums.copy_(torch.from_numpy(comb[:K * Dp].copy()))
        counts.copy_(torch.from_numpy(comb[K * Dp:K * Dp + K].copy()))
        self.finalize(C_old, sums, counts, C_new, shift, torch.tensor([nc], dtype=torch.float64), status, cnorm_new,
                      spherical=spherical)

    def l2norm_rows(self, X, out):
        x = _np(X).astype(np.float64)
        out.copy_(torch.from_numpy((x / np.sqrt((x * x).sum(1, keepdims=True))).astype(np.float32)))

    def finalize(self, C_old, sums, counts, C_new, shift, n_changed, status, cnorm_new=None, spherical=False):
        K, Dp = C_old.shape
        new, sh, tot = ok.finalize(_np(C_old), _np(sums).reshape(K, Dp), _np(counts))
        if spherical:
            nrm = np.sqrt((new.astype(np.float32) ** 2).sum(1, keepdims=True, dtype=np.float32))
            new = np.where(nrm > 0, new / np.where(nrm > 0, nrm, 1), new).astype(np.float32)
            d = (new - _np(C_old)).astype(np.float64)
            sh = np.sqrt((d * d).sum(1)).astype(np.float32)
            tot = float((sh.astype(np.float64) ** 2).sum())
        C_new.copy_(torch.from_numpy(new))
        if cnorm_new is not None:
            cnorm_new.copy_(torch.from_numpy(ok.row_sqnorm_chain(np.ascontiguousarray(new))))
        shift.copy_(torch.from_numpy(sh))
        status[0] = tot
        status[1] = float((_np(counts) == 0).sum())
        status[2] = float(n_changed.item()) if n_changed is not None else -1.0
        status[3] = 0.0

    def dist_to_assigned(self, X, C, labels, dist):
        dist.copy_(torch.from_numpy(ok.dist_to_assigned(_np(X), _np(C), _np(labels))))

    def sum_f64(self, v, out):
        x = _np(v).astype(np.float64)
        tot = 0.0
        for s in range(0, len(x), 256):
            a = 0.0
            for e in x[s:s + 256]:
                a += e
            tot += a
        out[0] = tot

    def select_far(self, dist, n_sel, far_idx, far_dist):
        d = _np(dist)
        N = len(d)
        for e in range(n_sel):
            i = int(np.argmax(d)) if N else N            # first max = lowest row on ties
            if d[i] < -1.5:
                far_idx[e], far_dist[e] = N, -1.0
                continue
            far_idx[e], far_dist[e] = i, float(d[i])
            d[i] = -2.0

    def apply_relocation(self, xfar, old_ids, new_ids, sums, counts):
        n, Dp = xfar.shape
        s = _np(sums).reshape(-1, Dp)
        c = _np(counts)
        for e in range(n):
            o, w = int(old_ids[e]), int(new_ids[e])
            s[o] -= _np(xfar)[e]
            s[w] = _np(xfar)[e]
            c[w] = 1.0
            c[o] -= 1.0

    def kpp_step(self, X, cand, T, closest, newdist, pot):
        x = _np(X).astype(np.float64)
        for t in range(T):
            d = ((x - x[int(cand[t])]) ** 2).sum(1).astype(np.float32)
            if closest is not None:
                d = np.minimum(d, _np(closest))
            if newdist.dim() == 1:
                newdist.copy_(torch.from_numpy(d))
            else:
                newdist[t].copy_(torch.from_numpy(d))
            pot[t] = float(d.astype(np.float64).sum())

    def cumsum_search(self, v, vals, T, idx_out):
        cs = np.cumsum(_np(v).astype(np.float64))
        r = np.clip(np.searchsorted(cs, _np(vals)[:T]), None, len(cs) - 1)
        idx_out[:T] = torch.from_numpy(r.astype(np.int32))
