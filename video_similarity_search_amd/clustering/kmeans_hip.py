"""
Host driver of the gfx950 k-means (C ABI: include/slic_hip.h, kernels: csrc/kmeans.hip).

Mirrors the object the reference builds at clustering/cluster_masks.py:70-71,
    KMeans(n_clusters=k, n_init=10).fit(embeddings).labels_
i.e. sklearn.cluster.KMeans (algorithm='lloyd'):
    KMeans.fit             sklearn/cluster/_kmeans.py:1427-1540  (centre X, tol, n_init loop, best inertia)
    _kmeans_single_lloyd   sklearn/cluster/_kmeans.py:624-752    (loop, strict / tol convergence, final E-step)
    _kmeans_plusplus       sklearn/cluster/_kmeans.py:174-277    (seeding; host RNG, device distances)
Only control flow lives here; every reduction/contraction is a HIP kernel reached through
`HipKernels` (one method per C-ABI entry point).  One host sync per Lloyd iteration (a 4-double
status word).

Multi-GPU: pass `process_group`; X is then this rank's contiguous row shard
(rank order == row order).  An iteration is two foreign calls around ONE collective over RCCL/xGMI
(it replaces the rank-0 k-means + barrier of online_train.py:625-662):
    slic_kmeans_lloyd_local  : E-step + ordered M-step on the rank's rows -> payload [K*D sums | K counts | n_changed]
    exchange='allreduce' (default): ONE all-reduce(sum) of the payload, widened to fp64 — a sum of a few fp32 values in
        fp64 is exact, so the result does not depend on RCCL's reduction order and every rank holds bit-identical
        centres; equals the oracle run with n_shards = -world (fp64 combine);
    exchange='allgather': ONE all-gather of the fp32 payloads, added in rank order on every GPU; equals the oracle
        run with n_shards = world (== sklearn's per-thread buffers reduced in thread order);
    slic_kmeans_lloyd_global : combine + averaging + shift + next norms + status word.
"""
import ctypes
import os
import time

import numpy as np
import torch

from .. import _lib
from .._lib import call, ptr, stream


_POISON = float(2 ** 40)      # in the n_changed high slot of a sharded payload: this rank's local half failed (exact in fp32 and fp64, sums of <= 16 too)

class HipKernels:
    """Thin argument marshalling for the k-means entry points of libslic_hip.so.  The only kernel
    provider the package ships; `KMeans(kernels=...)` exists so the multi-process control flow can be
    exercised under gloo on a GPU-less host by the tests (tests/kmeans_cpu_kernels.py)."""

    device_type = "cuda"

    def check(self):
        if not torch.cuda.is_available():
            raise _lib.SlicError("k-means needs a gfx950 device: libslic_hip.so has no CPU path")
        _lib.check(_lib.load().slic_device_check(), "slic_device_check")

    def to_device(self, X):
        if not torch.is_tensor(X):
            X = torch.as_tensor(np.ascontiguousarray(X, dtype=np.float32))
        return X.detach().to(device="cuda", dtype=torch.float32).contiguous()

    def col_stats(self, X):
        N, Dp = X.shape
        cs = torch.empty(2, Dp, dtype=torch.float64, device=X.device)
        ws = _lib.workspace(_lib.load().slic_col_stats_workspace_bytes(N, Dp), X.device, "km_cs")
        call("slic_col_stats", ptr(X), N, Dp, X.stride(0), ptr(cs[0]), ptr(cs[1]), ptr(ws), stream())
        return cs

    def sub_rowvec(self, X, v, out):
        call("slic_sub_rowvec", ptr(X), X.shape[0], X.shape[1], X.stride(0), ptr(v), ptr(out), out.stride(0), stream())

    def cnorm(self, C, cnorm):
        call("slic_kmeans_cnorm", ptr(C), C.shape[0], C.shape[1], C.stride(0), ptr(cnorm), stream())

    def assign(self, X, C, cnorm, labels, labels_old, n_changed):
        N, Dp = X.shape
        K = C.shape[0]
        ws = _lib.workspace(_lib.load().slic_kmeans_assign_workspace_bytes(N, K), X.device, "km_assign")
        call("slic_kmeans_assign", ptr(X), N, Dp, X.stride(0), ptr(C), K, C.stride(0), ptr(cnorm), ptr(labels),
             ptr(labels_old), ptr(n_changed), None, ptr(ws), stream())

    uses_perm = True       # E-step on k-permuted copies of X and the centres (LDS-DMA kernel)

    def permute_k8(self, X, Xp):
        call("slic_kmeans_permute_k8", ptr(X), X.shape[0], X.shape[1], X.stride(0), ptr(Xp), Xp.stride(0), stream())

    def assign_perm(self, Xp, Cp, cnorm, labels, labels_old, n_changed):
        N, Dp = Xp.shape
        K = Cp.shape[0]
        ws = _lib.workspace(_lib.load().slic_kmeans_assign_workspace_bytes(N, K), Xp.device, "km_assign")
        call("slic_kmeans_assign_perm", ptr(Xp), N, Dp, Xp.stride(0), ptr(Cp), K, Cp.stride(0), ptr(cnorm), ptr(labels),
             ptr(labels_old), ptr(n_changed), None, ptr(ws), stream())

    def lloyd_step(self, X, Xp, C_old, Cp_old, cnorm_old, labels, labels_old, n_changed, sums, counts, C_new, Cp_new,
                   cnorm_new, shift, status, spherical=False):
        """one whole unsharded iteration in one foreign call (the Python loop is otherwise the bottleneck at ~0.6 ms)"""
        N, Dp = X.shape
        K = C_old.shape[0]
        assert X.stride(0) == Xp.stride(0) and C_old.is_contiguous() and Cp_old.is_contiguous() and C_new.is_contiguous()
        ws = _lib.workspace(_lib.load().slic_kmeans_lloyd_step_workspace_bytes(N, K), X.device, "km_step")
        call("slic_kmeans_lloyd_step", ptr(X), ptr(Xp), N, Dp, X.stride(0), ptr(C_old), ptr(Cp_old), ptr(cnorm_old), K,
             ptr(labels), ptr(labels_old), ptr(n_changed), ptr(sums), ptr(counts), ptr(C_new), ptr(Cp_new), ptr(cnorm_new),
             ptr(shift), int(spherical), ptr(status), ptr(ws), stream())

    def lloyd_local(self, X, Xp, C_old, Cp_old, cnorm_old, labels, labels_old, payload):
        """sharded iteration, part 1 (before the collective): E-step + ordered M-step -> payload (fp32 or fp64 tensor of
        K*D + K + 2 numbers)"""
        N, Dp = X.shape
        K = Cp_old.shape[0]
        assert X.stride(0) == Xp.stride(0) and Cp_old.is_contiguous() and payload.numel() == K * Dp + K + 2
        ws = _lib.workspace(_lib.load().slic_kmeans_lloyd_local_workspace_bytes(N, K), X.device, "km_step")
        call("slic_kmeans_lloyd_local", ptr(X), ptr(Xp), N, Dp, X.stride(0), ptr(Cp_old), ptr(cnorm_old), K, ptr(labels),
             ptr(labels_old), ptr(payload), int(payload.dtype == torch.float64), ptr(ws), stream())

    def lloyd_global(self, parts, C_old, sums, counts, C_new, Cp_new, cnorm_new, shift, status, spherical=False):
        """sharded iteration, part 2 (after the collective): parts = [W, K*D + K + 2] gathered fp32 payloads or
        [1, K*D + K + 2] reduced fp64 payload -> combined sums / counts, new centres, status word"""
        K, Dp = C_old.shape
        W, stride = parts.shape
        call("slic_kmeans_lloyd_global", ptr(parts), int(parts.dtype == torch.float64), stride, W, ptr(C_old), K, Dp,
             ptr(sums), ptr(counts), ptr(C_new), ptr(Cp_new), ptr(cnorm_new), ptr(shift), int(spherical), ptr(status), stream())

    def l2norm_rows(self, X, out):
        call("slic_l2norm_rows", ptr(X), X.shape[0], X.shape[1], X.stride(0), ptr(out), out.stride(0), stream())

    def kpp_run(self, X, first, K, T, uniforms, idx_out, Xp=None, xnorm=None):
        N, Dp = X.shape
        assert Xp is None or Xp.stride(0) == X.stride(0)
        ws = _lib.workspace(_lib.load().slic_kmeanspp_run_workspace_bytes(N, T), X.device, "kpp_run")
        call("slic_kmeanspp_run", ptr(X), N, Dp, X.stride(0), int(first), int(K), int(T), ptr(uniforms), ptr(idx_out),
             ptr(Xp), ptr(xnorm), ptr(ws), stream())

    def kpp_run_batch(self, Xp, xnorm, firsts, K, T, uniforms, idx_out):
        """all R = len(firsts) initialisations of an n_init loop in lock-step (slic_kmeanspp_run_batch)"""
        import ctypes
        N, Dp = Xp.shape
        R = len(firsts)
        arr = (ctypes.c_int32 * R)(*[int(f) for f in firsts])
        ws = _lib.workspace(_lib.load().slic_kmeanspp_run_batch_workspace_bytes(N, T, R), Xp.device, "kpp_batch")
        call("slic_kmeanspp_run_batch", ptr(Xp), ptr(xnorm), N, Dp, Xp.stride(0), R, arr, int(K), int(T), ptr(uniforms), ptr(idx_out),
             ptr(ws), stream())

    def accumulate(self, X, labels, K, sums, counts):
        N, Dp = X.shape
        ws = _lib.workspace(_lib.load().slic_kmeans_accumulate_workspace_bytes(N, K), X.device, "km_accum")
        call("slic_kmeans_accumulate", ptr(X), N, Dp, X.stride(0), ptr(labels), K, ptr(sums), ptr(counts),
             ptr(ws), stream())

    def combine_shards(self, allpart, K, Dp, sums, counts):
        W, stride = allpart.shape
        call("slic_kmeans_combine_shards", ptr(allpart), ptr(allpart[0, K * Dp:]), stride, W, K, Dp,
             ptr(sums), ptr(counts), stream())

    def finalize(self, C_old, sums, counts, C_new, shift, n_changed, status, cnorm_new=None, C_new_perm=None, spherical=False):
        K, Dp = C_old.shape
        call("slic_kmeans_finalize", ptr(C_old), ptr(sums), ptr(counts), K, Dp, ptr(C_new), ptr(shift),
             ptr(cnorm_new), ptr(C_new_perm), int(spherical), ptr(n_changed), ptr(status), stream())

    def dist_to_assigned(self, X, C, labels, dist):
        call("slic_kmeans_dist_to_assigned", ptr(X), X.shape[0], X.shape[1], X.stride(0), ptr(C), C.stride(0),
             ptr(labels), ptr(dist), stream())

    def sum_f64(self, v, out):
        ws = _lib.workspace(_lib.load().slic_sum_f32_to_f64_workspace_bytes(v.numel()), v.device, "km_sum")
        call("slic_sum_f32_to_f64", ptr(v), v.numel(), ptr(out), ptr(ws), stream())

    def select_far(self, dist, n_sel, far_idx, far_dist):
        call("slic_kmeans_select_far", ptr(dist), dist.numel(), n_sel, ptr(far_idx), ptr(far_dist), stream())

    def apply_relocation(self, xfar, old_ids, new_ids, sums, counts):
        n, Dp = xfar.shape
        call("slic_kmeans_apply_relocation", ptr(xfar), xfar.stride(0), ptr(old_ids), ptr(new_ids), n, Dp,
             ptr(sums), ptr(counts), stream())

    def kpp_step(self, X, cand, T, closest, newdist, pot):
        N, Dp = X.shape
        ws = _lib.workspace(_lib.load().slic_kmeanspp_step_workspace_bytes(N, T), X.device, "kpp_p")
        call("slic_kmeanspp_step", ptr(X), N, Dp, X.stride(0), ptr(cand), T, ptr(closest), ptr(newdist), ptr(pot),
             ptr(ws), stream())

    def cumsum_search(self, v, vals, T, idx_out):
        ws = _lib.workspace(_lib.load().slic_cumsum_search_workspace_bytes(v.numel()), v.device, "kpp_c")
        call("slic_cumsum_search", ptr(v), v.numel(), ptr(vals), T, ptr(idx_out), ptr(ws), stream())


# process group -> slic_comm* (an RCCL communicator of the library's own, created once per group).  Keyed on the group OBJECT through
# weak references: a destroyed group's id() may be re-used by a new group with other members, and a stale communicator would hang or
# reduce over the wrong ranks; when the group goes away its communicator is destroyed with it.
import weakref

_COMMS = weakref.WeakKeyDictionary()


_EXITING = []


def _mark_exiting():
    _EXITING.append(True)


class _CommHandle:
    """owns one slic_comm*: slic_comm_destroy (finalize, bounded settle, destroy) when the handle dies because its group was collected;
    slic_comm_abort at interpreter exit — torch may have torn its process group down by then and a peer may be gone: the exit path
    must not wait for anybody"""

    _hooked = False

    def __init__(self, ptr_):
        self.ptr = ptr_
        self._fin = weakref.finalize(self, _CommHandle._close, ptr_)
        if not _CommHandle._hooked:
            # registered AFTER weakref's own exit hook (created with the first finalize object), so it runs BEFORE it (LIFO)
            import atexit
            atexit.register(_mark_exiting)
            _CommHandle._hooked = True

    @staticmethod
    def _close(ptr_):
        try:
            lib = _lib.load()
            (lib.slic_comm_abort if _EXITING else lib.slic_comm_destroy)(ptr_)
        except Exception:
            pass


def comm_timeout_ms():
    """deadline of the library communicator's waits (SLIC_COMM_TIMEOUT_MS, default five minutes — torch.distributed's own collectives
    default to ten): a peer that never joins makes every rank raise instead of hang"""
    return int(os.environ.get("SLIC_COMM_TIMEOUT_MS", "300000"))


def _slic_comm(pg, dev):
    """the C-ABI communicator for the sharded iteration's all-reduce (include/slic_hip.h: slic_comm_create_timeout): rank 0 of the
    group draws the unique id, torch.distributed carries it to the others, every rank joins on its device — under a deadline"""
    import ctypes
    h = _COMMS.get(pg)
    if h is not None:
        return h.ptr
    lib = _lib.load()
    W, rank = torch.distributed.get_world_size(pg), torch.distributed.get_rank(pg)
    idt = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        buf = (ctypes.c_ubyte * 128)()
        _lib.check(lib.slic_comm_unique_id(buf), "slic_comm_unique_id")
        idt.copy_(torch.tensor(list(buf), dtype=torch.uint8))
    torch.distributed.broadcast(idt, src=torch.distributed.get_global_rank(pg, 0), group=pg)
    raw = bytes(idt.cpu().numpy().tobytes())
    comm = ctypes.c_void_p()
    _lib.check(lib.slic_comm_create_timeout(raw, W, rank, comm_timeout_ms(), ctypes.byref(comm)), "slic_comm_create_timeout")
    _COMMS[pg] = _CommHandle(comm)
    return comm


_ONESHOTS = weakref.WeakKeyDictionary()


class _OneshotHandle:
    """owns one slic_oneshot*; destroyed with its process group or at interpreter exit (slic_oneshot_destroy never waits for a peer)"""

    def __init__(self, ptr_, max_n):
        self.ptr, self.max_n = ptr_, max_n
        self._fin = weakref.finalize(self, _OneshotHandle._close, ptr_)

    @staticmethod
    def _close(ptr_):
        try:
            _lib.load().slic_oneshot_destroy(ptr_)
        except Exception:
            pass


def _pg_all_gather_bytes(pg, raw, dev):
    """all-gather one small byte string per rank over the process group (RCCL: device tensors; gloo: host tensors) -> bytes, rank order"""
    W = torch.distributed.get_world_size(pg)
    on_dev = torch.distributed.get_backend(pg) == "nccl"
    t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).clone()
    t = t.to(dev) if on_dev else t
    out = torch.empty(W * t.numel(), dtype=torch.uint8, device=t.device)
    torch.distributed.all_gather_into_tensor(out, t, group=pg)
    return bytes(out.cpu().numpy().tobytes())


def _slic_oneshot(pg, dev, n):
    """the one-shot exchange of the sharded iteration (include/slic_hip.h: slic_oneshot_*): every rank allocates its inbox, the IPC handles
    travel over the process group, every rank maps its peers.  One communicator per group, re-made when a larger payload comes along."""
    h = _ONESHOTS.get(pg)
    if h is not None and h.max_n >= n:
        return h.ptr
    lib = _lib.load()
    W, rank = torch.distributed.get_world_size(pg), torch.distributed.get_rank(pg)
    if W > 1 and os.environ.get("SLIC_ONESHOT_MULTI_GPU", "0") != "1":
        # EXPERIMENTAL between GPUs: every test of this exchange ran its ranks as processes on ONE device (one-GPU boxes); a run over xGMI has
        # never been compared with the all-reduce route.  Ranks on different devices must opt in.  Every rank sees the same
        # gathered identities, so every rank raises (or none does).
        import socket
        pr = torch.cuda.get_device_properties(dev)
        ident = "%s/%s/%s/%s/%s" % (socket.gethostname(), getattr(pr, "uuid", ""), getattr(pr, "pci_domain_id", ""), getattr(pr, "pci_bus_id", ""),
                                    getattr(pr, "pci_device_id", ""))
        ids = _pg_all_gather_bytes(pg, ident.encode()[:96].ljust(96, b"\0"), dev)
        if len({ids[i * 96:(i + 1) * 96] for i in range(W)}) > 1:
            raise _lib.SlicError("exchange='oneshot' between different GPUs is experimental (never verified against the RCCL all-reduce on a multi-GPU "
                                 "node): set SLIC_ONESHOT_MULTI_GPU=1 to opt in, or use exchange='allreduce'")
    hb = 64
    mine = (ctypes.c_ubyte * hb)()
    comm = ctypes.c_void_p()
    with torch.cuda.device(dev):
        _lib.check(lib.slic_oneshot_create(W, rank, n, comm_timeout_ms(), ctypes.byref(comm), mine), "slic_oneshot_create")
        allh = _pg_all_gather_bytes(pg, bytes(mine), dev)
        _lib.check(lib.slic_oneshot_connect(comm, allh), "slic_oneshot_connect")
    # nobody may push into an inbox before its owner has mapped... (a push only needs the PUSHER's mapping; the barrier keeps a fast rank's
    # first exchange from racing a slow rank's connect bookkeeping and gives create / connect failures a common point to surface)
    torch.distributed.barrier(group=pg)
    _ONESHOTS[pg] = _OneshotHandle(comm, n)
    return comm


def _dist_on(pg):
    # a process group of ONE rank still takes the sharded path (all-gather of one partial, ordered add): that is
    # how a single-GPU box exercises the RCCL code path
    return pg is not None and torch.distributed.is_initialized()


class KMeans:
    """sklearn-shaped: KMeans(n_clusters, n_init=10, max_iter=300, tol=1e-4).fit(X) -> labels_, cluster_centers_,
    inertia_, n_iter_.  `init` may be 'k-means++' (default, as the reference uses), an [K, D] array, or a list of
    such arrays (one per run).  X: torch tensor [N, D] fp32 on a gfx950 device (a CPU tensor / ndarray is
    copied to the current device)."""

    def __init__(self, n_clusters, n_init=10, max_iter=300, tol=1e-4, init="k-means++", random_state=None,
                 process_group=None, fixed_iters=False, trace=False, kernels=None, spherical=False, exchange=None):
        self.n_clusters = int(n_clusters)
        self.n_init = int(n_init)
        self.max_iter = int(max_iter)
        self.tol = float(tol)
        self.init = init
        self.random_state = random_state
        self.process_group = process_group
        self.fixed_iters = bool(fixed_iters)   # throughput runs: skip the stopping tests
        self.trace = bool(trace)               # keep every iteration's labels (tests)
        # spherical k-means (clustering/cluster_masks.py:73-77 -> spherecluster.SphericalKMeans): rows L2-normalised, no
        # mean-centring, centres renormalised after every averaging, tol compared unscaled
        self.spherical = bool(spherical)
        exchange = exchange or os.environ.get("SLIC_KMEANS_EXCHANGE", "allreduce")
        assert exchange in ("allreduce", "allgather", "oneshot"), exchange
        self.exchange = exchange               # the sharded run's one collective per iteration (module docstring)
        self.k = kernels if kernels is not None else HipKernels()

    # ------------------------------------------------------------------ helpers
    def _rng(self):
        """check_random_state(self.random_state), ONCE per fit (KMeans.fit, _kmeans.py:1467: the n_init runs share the stream)"""
        rs = self.random_state
        if rs is None:
            return np.random.mtrand._rand     # sklearn check_random_state(None): numpy's global RandomState
        if isinstance(rs, (int, np.integer)):
            if getattr(self, "_fit_rs", None) is None:
                self._fit_rs = np.random.RandomState(rs)
            return self._fit_rs
        return rs

    def _host_staged(self, t):
        """device tensors over a gloo group (two ranks sharing ONE GPU — how a one-GPU box runs exchange='oneshot' with two processes; RCCL
        cannot put two ranks on a device) go through host copies; RCCL groups and host tensors are used as they are"""
        return t.is_cuda and torch.distributed.get_backend(self.process_group) == "gloo"

    def _gather(self, t):
        """all-gather a per-rank tensor [*] -> [W, *] (same shape on every rank)"""
        W = torch.distributed.get_world_size(self.process_group)
        src = t.contiguous().reshape(-1)
        if self._host_staged(t):
            src = src.cpu()
        flat = torch.empty(W * t.numel(), dtype=t.dtype, device=src.device)    # concatenated form (gloo and RCCL)
        torch.distributed.all_gather_into_tensor(flat, src, group=self.process_group)
        return flat.to(t.device).view((W,) + tuple(t.shape))

    def _col_stats(self, X):
        """global column sum / sum of squares as float64 numpy (rank partials added in rank order)"""
        cs = self.k.col_stats(X)
        if self._sharded:
            allcs = self._gather(cs).cpu().numpy()      # [W, 2, Dp]
            tot = np.zeros_like(allcs[0])
            for r in range(allcs.shape[0]):
                tot += allcs[r]
            return tot
        return cs.cpu().numpy()

    # ------------------------------------------------------------------ fit
    def fit(self, X):
        self._fit_rs = None
        self.init_indices_log_ = []
        self.k.check()
        X = self.k.to_device(X)
        N, D = X.shape
        K = self.n_clusters
        self._sharded = _dist_on(self.process_group)
        dev = X.device
        # pad the feature dim to a multiple of 8 with zero columns (they add exact zeros to every chain)
        Dp = (D + 7) // 8 * 8
        if Dp != D:
            Xp = torch.zeros(N, Dp, dtype=torch.float32, device=dev)
            Xp[:, :D] = X
            X = Xp
        if self._sharded:
            sizes = self._gather(torch.tensor([N], dtype=torch.int64, device=dev)).cpu().numpy().reshape(-1)
            rank = torch.distributed.get_rank(self.process_group)
            self._row0 = int(sizes[:rank].sum())
            self._sizes = sizes
            Ng = int(sizes.sum())
        else:
            self._row0, self._sizes, Ng = 0, np.array([N]), N

        if self.spherical:
            # SphericalKMeans.fit: X = normalize(X); the data is NOT centred and tol is used as given
            mean = np.zeros(Dp, np.float32)
            Xc = torch.empty_like(X)
            self.k.l2norm_rows(X, Xc)
            tol_abs = self.tol
        else:
            # X -= X.mean(axis=0)   (_kmeans.py:1479-1481)
            cs = self._col_stats(X)
            mean = (cs[0] / float(Ng)).astype(np.float32)
            mean_d = torch.from_numpy(mean).to(dev)
            Xc = torch.empty_like(X)
            self.k.sub_rowvec(X, mean_d, Xc)
        # tol = mean(var(Xc, axis=0)) * tol   (_tolerance, _kmeans.py:279-288)
        if self.spherical:
            pass
        elif self.tol == 0:
            tol_abs = 0.0
        else:
            cs2 = self._col_stats(Xc)
            m = cs2[0] / float(Ng)
            var = cs2[1] / float(Ng) - m * m
            tol_abs = float(var[:D].sum() / D) * self.tol
        self.tol_abs_ = tol_abs

        inits = self.init
        if isinstance(inits, str):
            assert inits == "k-means++", inits
            inits = None
        elif isinstance(inits, (list, tuple)):
            inits = [np.asarray(a, np.float32) for a in inits]
        else:
            a = inits.detach().cpu().numpy() if torch.is_tensor(inits) else np.asarray(inits)
            inits = [np.asarray(a, np.float32)]
        n_runs = self.n_init if inits is None else len(inits)

        best = None
        seeded = self._kmeans_plusplus_all(Xc, K, n_runs) if (inits is None and n_runs > 1) else None
        for run in range(n_runs):
            if seeded is not None:
                C0 = seeded[run]
            elif inits is None:
                C0 = self._kmeans_plusplus(Xc, K)          # rows of the centred matrix
            else:
                c = np.zeros((K, Dp), np.float32)
                c[:, :D] = inits[run] - mean[None, :D]
                C0 = torch.from_numpy(c).to(dev)
            res = self._lloyd_single(Xc, C0, tol_abs)
            if best is None or res["inertia"] < best["inertia"]:
                best = res
        self.labels_ = best["labels"].cpu().numpy()                 # np.int32, like sklearn's labels_
        self.labels_device_ = best["labels"]
        self.cluster_centers_ = best["centers"].cpu().numpy()[:, :D] + mean[None, :D]
        self.inertia_ = best["inertia"]
        self.n_iter_ = best["n_iter"]
        self.strict_ = best["strict"]
        self.n_relocations_ = best["n_relocations"]
        self.trace_ = best.get("trace")
        return self

    # ------------------------------------------------------------------ one Lloyd run
    def _lloyd_single(self, Xc, C, tol_abs):
        """_kmeans_single_lloyd with the host loop running ONE iteration behind the device: iteration it+1 is enqueued
        before iteration it's 4-double status word is read, so the read-back (and this Python) hide under the next
        E-step.  A speculatively launched iteration is simply ignored when the previous one turns out to have converged
        (it only reads the state it would need to keep: rings of 3 centre / label buffers, 2 partial-sum buffers)."""
        k = self.k
        N, Dp = Xc.shape
        K = C.shape[0]
        dev = Xc.device
        on_gpu = Xc.is_cuda
        Cb = [C.contiguous().clone(), torch.empty_like(C), torch.empty_like(C)]          # iteration it: Cb[it%3] -> Cb[(it+1)%3]
        Lb = [torch.full((N,), -1, dtype=torch.int32, device=dev) for _ in range(3)]      # labels of iteration it: Lb[it%3]
        cnorm = [torch.empty(K, dtype=torch.float32, device=dev) for _ in range(3)]        # norms of Cb[i]: written by finalize
        perm = bool(getattr(k, "uses_perm", False))
        sph = dict(spherical=True) if self.spherical else {}
        if perm:
            Xp = self._permuted(Xc)                                                        # once per fit, shared by the inits
            Cp = [torch.empty_like(C) for _ in range(3)]                                   # permuted twins of Cb
        else:
            Cp = [None] * 3
        n_changed = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(2)]
        part = [torch.empty(K * Dp + K, dtype=torch.float32, device=dev) for _ in range(2)]   # [sums | counts]: the all-gather unit
        shift = torch.empty(K, dtype=torch.float32, device=dev)
        status = [torch.empty(4, dtype=torch.float64, device=dev) for _ in range(2)]
        host = [torch.empty(4, dtype=torch.float64).pin_memory() if on_gpu else torch.empty(4, dtype=torch.float64) for _ in range(2)]
        ev = [torch.cuda.Event() if on_gpu else None for _ in range(2)]
        done = [torch.cuda.Event() if on_gpu else None for _ in range(2)]
        side = self._side_stream(dev) if on_gpu else None
        comm = None
        if self._sharded:
            W = torch.distributed.get_world_size(self.process_group)
            PL = K * Dp + K + 2                                          # [sums | counts | n_changed lo, hi]
            if self.exchange in ("allreduce", "oneshot"):
                payload = [torch.empty(1, PL, dtype=torch.float64, device=dev) for _ in range(2)]
                parts = payload                                          # reduced in place
            else:
                payload = [torch.empty(1, PL, dtype=torch.float32, device=dev) for _ in range(2)]
                parts = [torch.empty(W, PL, dtype=torch.float32, device=dev) for _ in range(2)]
            gsums = [torch.empty(K * Dp, dtype=torch.float32, device=dev) for _ in range(2)]
            gcounts = [torch.empty(K, dtype=torch.float32, device=dev) for _ in range(2)]
            # the all-reduce goes through torch.distributed's process group (RCCL on the GPU, gloo in the CPU tests): its
            # collectives carry the group's timeout and it is the path every multi-rank test runs.  SLIC_KMEANS_COMM=slic opts
            # into the library's own RCCL communicator (slic_allreduce_f64 on the compute stream, the C ABI's collective for
            # callers without torch.distributed); its waits are bounded too (slic_comm_create_timeout / slic_comm_wait)
            if (self.exchange == "allreduce" and on_gpu and torch.distributed.get_backend(self.process_group) == "nccl"
                    and os.environ.get("SLIC_KMEANS_COMM", "torch") == "slic"):
                comm = _slic_comm(self.process_group, dev)
            # exchange="oneshot": the library's one-shot all-to-all over peer-mapped memory (slic_allreduce_oneshot_f64: one kernel per
            # iteration instead of a ring all-reduce; csrc/oneshot.hip).  The process group only carries the IPC handles at set-up.
            oneshot = None
            if self.exchange == "oneshot":
                assert on_gpu, "exchange='oneshot' moves device memory between GPUs"
                oneshot = _slic_oneshot(self.process_group, dev, PL)
            # what a benchmark line reports about the iteration's one exchange (bench.py: secondary.exchange)
            self.communicator_kind_ = ("slic_oneshot (peer-mapped inboxes, one kernel per exchange; no RCCL in the loop)" if oneshot is not None else
                                       "slic_comm (the library's own RCCL communicator, slic_allreduce_f64)" if comm is not None else
                                       f"torch.distributed process group ({torch.distributed.get_backend(self.process_group)})")
            self.payload_bytes_ = int(payload[0].numel() * payload[0].element_size())
        else:
            gsums = [p[: K * Dp] for p in part]
            gcounts = [p[K * Dp:] for p in part]

        def read_back(sl):
            if on_gpu:
                # the 32-byte status read-back goes through a side stream: on the compute stream the D2H copy and its
                # system-scope release would hold the next iteration's first kernel back by ~35 us
                done[sl].record()
                side.wait_event(done[sl])
                with torch.cuda.stream(side):
                    host[sl].copy_(status[sl], non_blocking=True)
                ev[sl].record(side)
            else:
                host[sl].copy_(status[sl])

        local_failure = []

        def launch(it):
            """enqueue iteration `it` (E-step, M-step, averaging) and the async read-back of its status word"""
            sl = it & 1
            Cin, Cout = Cb[it % 3], Cb[(it + 1) % 3]
            lab, lab_old = Lb[it % 3], Lb[(it + 2) % 3]            # (it - 1) % 3
            if it == 0:
                k.cnorm(Cin, cnorm[0])                             # later norms / permuted centres come out of finalize
                if perm:
                    k.permute_k8(Cin, Cp[0])
            if perm and not self._sharded and hasattr(k, "lloyd_step"):
                k.lloyd_step(Xc, Xp, Cin, Cp[it % 3], cnorm[it % 3], lab, lab_old, n_changed[sl], gsums[sl], gcounts[sl],
                             Cout, Cp[(it + 1) % 3], cnorm[(it + 1) % 3], shift, status[sl], **sph)
                read_back(sl)
                return
            if self._sharded:
                # two foreign calls around the iteration's ONE collective.  A rank whose local half raises (an allocation that fails, a bad
                # argument) must still ENTER the collective, or its peers wait for it until the process group's timeout: it contributes a
                # POISONED payload — the n_changed high slot, which a healthy rank fills with < 2^11 — from then on, and raises the cause
                # where the loop reads that iteration's status; its peers see the poison in the same status word and raise there too.
                if not local_failure:
                    try:
                        k.lloyd_local(Xc, Xp if perm else None, Cin, Cp[it % 3], cnorm[it % 3], lab, lab_old, payload[sl])
                    except Exception as e:                              # noqa: BLE001 — re-raised by read() of this iteration
                        local_failure.append((it, e))
                if local_failure:
                    payload[sl].zero_()
                    payload[sl].view(-1)[-1] = _POISON
                if self.exchange == "oneshot":
                    call("slic_allreduce_oneshot_f64", oneshot, ptr(payload[sl]), payload[sl].numel(), stream())
                elif self.exchange == "allreduce" and comm is not None:
                    call("slic_allreduce_f64", comm, ptr(payload[sl]), payload[sl].numel(), stream())
                elif self._host_staged(payload[sl]):
                    if self.exchange == "allreduce":
                        h = payload[sl].cpu()
                        torch.distributed.all_reduce(h, group=self.process_group)
                        payload[sl].copy_(h)
                    else:
                        parts[sl].copy_(self._gather(payload[sl].view(-1)).view_as(parts[sl]))
                elif self.exchange == "allreduce":
                    torch.distributed.all_reduce(payload[sl], group=self.process_group)
                else:
                    torch.distributed.all_gather_into_tensor(parts[sl].view(-1), payload[sl].view(-1), group=self.process_group)
                if local_failure:
                    return          # no global half, no read-back: read() of the failed iteration raises (the loop runs one launch ahead of
                                    # its reads, so this rank keeps entering the collectives its peers enter until they all get there)
                k.lloyd_global(parts[sl], Cin, gsums[sl], gcounts[sl], Cout, Cp[(it + 1) % 3], cnorm[(it + 1) % 3], shift,
                               status[sl], **sph)
                read_back(sl)
                return
            n_changed[sl].zero_()
            if perm:
                k.assign_perm(Xp, Cp[it % 3], cnorm[it % 3], lab, lab_old, n_changed[sl])
            else:
                k.assign(Xc, Cin, cnorm[it % 3], lab, lab_old, n_changed[sl])
            k.accumulate(Xc, lab, K, part[sl][: K * Dp], part[sl][K * Dp:])
            k.finalize(Cin, gsums[sl], gcounts[sl], Cout, shift, n_changed[sl], status[sl], cnorm[(it + 1) % 3],
                       *((Cp[(it + 1) % 3],) if perm else ()), **sph)
            read_back(sl)

        def read(it):
            if local_failure and local_failure[0][0] <= it:
                raise local_failure[0][1]
            if on_gpu:
                if comm is not None:
                    # bounded: a peer that never joined the iteration's all-reduce aborts the communicator and raises here.  The wait is on
                    # THIS iteration's event (recorded behind its collective and status read-back): iteration it + 1, already enqueued,
                    # keeps running ahead, and the deadline covers one iteration
                    call("slic_comm_wait_event", comm, ctypes.c_void_p(ev[it & 1].cuda_event), comm_timeout_ms())
                ev[it & 1].synchronize()
                if self._sharded and self.exchange == "oneshot":
                    # the exchange kernel never hangs: a wait that ran out (a lost peer) is recorded and surfaces here
                    call("slic_oneshot_check", oneshot)
            st = host[it & 1].tolist()
            if self._sharded and st[2] >= _POISON:                      # n_changed = low + 2^20 * high: a peer's poisoned high slot
                raise _lib.SlicError("sharded k-means: a peer's local half of the Lloyd iteration failed (it raised the cause); "
                                     "every rank leaves the fit here instead of waiting in the next collective")
            return st

        trace = [] if self.trace else None
        strict = False
        n_reloc = 0
        it = 0
        if on_gpu:
            torch.cuda.synchronize(dev)
        t_loop = time.time()                                               # the Lloyd phase proper (SURVEY.md §8d metric 2)
        if self.max_iter > 0:
            launch(0)
        while it < self.max_iter:
            speculated = it + 1 < self.max_iter
            if speculated:
                launch(it + 1)
            shift_tot, n_empty, n_chg, _ = read(it)
            if n_empty > 0:
                # rare: _relocate_empty_clusters_dense on iteration it's sums, then redo its averaging — and the
                # speculative iteration, which ran on the un-relocated centres
                sl = it & 1
                if self._relocate(Xc, Cb[it % 3], Lb[it % 3], gsums[sl], gcounts[sl], int(n_empty)):
                    n_reloc += 1
                    k.finalize(Cb[it % 3], gsums[sl], gcounts[sl], Cb[(it + 1) % 3], shift, n_changed[sl], status[sl],
                               cnorm[(it + 1) % 3], *((Cp[(it + 1) % 3],) if perm else ()), **sph)
                    shift_tot = status[sl].cpu().tolist()[0]
                    if speculated:
                        launch(it + 1)
            if trace is not None:
                trace.append(Lb[it % 3].cpu().numpy().copy())
            if not self.fixed_iters:
                if n_chg == 0:                                         # np.array_equal(labels, labels_old)
                    strict = True
                    break
                if shift_tot <= tol_abs:
                    break
            it += 1
        n_iter = min(it + 1, self.max_iter)
        if on_gpu:
            torch.cuda.synchronize(dev)
        self.lloyd_seconds_, self.lloyd_iters_ = time.time() - t_loop, n_iter      # assign + update + convergence test
        last = n_iter - 1                                              # last executed (and kept) iteration
        C = Cb[(last + 1) % 3] if self.max_iter > 0 else Cb[0]
        labels = Lb[last % 3] if self.max_iter > 0 else Lb[0]
        if not strict:
            # rerun the E-step so labels match the final centres (_kmeans.py:736-748); any buffer but `C`'s reader state
            out = Lb[(last + 1) % 3]
            cn = cnorm[(last + 1) % 3] if self.max_iter > 0 else cnorm[0]
            if self.max_iter <= 0:
                k.cnorm(C, cn)
            if perm and self.max_iter > 0:
                k.assign_perm(Xp, Cp[(last + 1) % 3], cn, out, None, None)
            else:
                k.assign(Xc, C, cn, out, None, None)
            labels = out
        inertia = self._inertia(Xc, C, labels)
        res = dict(labels=labels, centers=C, inertia=inertia, n_iter=n_iter, strict=strict, n_relocations=n_reloc)
        if trace is not None:
            res["trace"] = np.stack(trace) if trace else np.zeros((0, N), np.int32)
        return res

    def _side_stream(self, dev):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=dev)
        return self._side

    def _row_norms(self, Xc):
        """squared row norms of the (centred) data, cached like the permuted copy"""
        key = (Xc.data_ptr(), tuple(Xc.shape), Xc._version)
        if getattr(self, "_norm_key", None) != key:
            xn = torch.empty(Xc.shape[0], dtype=torch.float32, device=Xc.device)
            self.k.cnorm(Xc, xn)
            self._norm_key, self._norm_X = key, xn
        return self._norm_X

    def _permuted(self, Xc):
        """k-permuted copy of the (centred) data for the LDS-DMA E-step; cached while Xc is the same tensor"""
        key = (Xc.data_ptr(), tuple(Xc.shape), Xc._version)
        if getattr(self, "_perm_key", None) != key:
            Xp = torch.empty_like(Xc)
            self.k.permute_k8(Xc, Xp)
            self._perm_key, self._perm_X = key, Xp
        return self._perm_X

    def _inertia(self, Xc, C, labels):
        N = Xc.shape[0]
        dist = torch.empty(N, dtype=torch.float32, device=Xc.device)
        out = torch.empty(1, dtype=torch.float64, device=Xc.device)
        self.k.dist_to_assigned(Xc, C, labels, dist)
        self.k.sum_f64(dist, out)
        if self._sharded:
            tot = 0.0
            for v in self._gather(out).cpu().numpy().reshape(-1):
                tot += float(v)
            return tot
        return float(out.item())

    def _relocate(self, Xc, C_old, labels, gsums, gcounts, n_empty):
        """_relocate_empty_clusters_dense (_k_means_common.pyx:167-211).  Returns False when max(dist) == 0."""
        k = self.k
        N, Dp = Xc.shape
        dev = Xc.device
        n_sel = n_empty
        dist = torch.empty(N, dtype=torch.float32, device=dev)
        k.dist_to_assigned(Xc, C_old, labels, dist)
        far_idx = torch.empty(n_sel, dtype=torch.int32, device=dev)
        far_dist = torch.empty(n_sel, dtype=torch.float32, device=dev)
        k.select_far(dist, n_sel, far_idx, far_dist)
        fi = far_idx.long()
        valid = fi < N                                   # a shard smaller than n_empty runs out of rows
        fi = fi.clamp(max=N - 1)
        xfar = Xc.index_select(0, fi)
        old_ids = labels.index_select(0, fi)
        fd = torch.where(valid, far_dist, torch.full_like(far_dist, -1.0))
        if self._sharded:
            # merge the per-rank candidates by (dist desc, global row asc), keep n_empty
            gidx = fi + self._row0
            fdh = self._gather(fd).reshape(-1).cpu().numpy()
            gih = self._gather(gidx).reshape(-1).cpu().numpy()
            allx = self._gather(xfar).reshape(-1, Dp)
            allold = self._gather(old_ids).reshape(-1)
            order = np.lexsort((gih, -fdh))[:n_empty]
            order = order[fdh[order] >= 0]
            sel = torch.from_numpy(order).to(dev)
            xfar = allx.index_select(0, sel).contiguous()
            old_ids = allold.index_select(0, sel).contiguous()
            fdh = fdh[order]
        else:
            fdh = fd.cpu().numpy()
        if len(fdh) == 0 or fdh[0] == 0.0:
            return False
        empty = np.nonzero(gcounts.cpu().numpy() == 0)[0].astype(np.int32)[: len(fdh)]
        new_ids = torch.from_numpy(empty).to(dev)
        k.apply_relocation(xfar[: len(empty)].contiguous(), old_ids[: len(empty)].contiguous(), new_ids, gsums, gcounts)
        return True

    # ------------------------------------------------------------------ k-means++
    def _kmeans_plusplus_all(self, Xc, K, R):
        """the k-means++ seeding of ALL R initialisations of the n_init loop at once, or None when the lock-step kernel does not
        apply (then the loop seeds run by run).  sklearn's loop draws, per run, the first row and then (K - 1) x T uniforms from
        ONE RNG stream; neither depends on the data, and Lloyd draws nothing, so the draws are made here in that order and the R
        runs — independent of each other from then on — take each step as one pass over X (slic_kmeanspp_run_batch)."""
        k = self.k
        T = 2 + int(np.log(K))
        if (self._sharded or not hasattr(k, "kpp_run_batch") or not getattr(k, "uses_perm", False) or T > 16 or R * T > 160
                or os.environ.get("SLIC_KPP_BATCH", "1") == "0"):
            return None
        N, Dp = Xc.shape
        if N * Xc.stride(0) * 4 >= (1 << 31):
            return None
        dev = Xc.device
        rs = self._rng()
        firsts, us = [], []
        for _ in range(R):
            firsts.append(int(rs.choice(N, p=np.full(N, 1.0 / N))))
            us.append(rs.uniform(size=(K - 1, T)) if K > 1 else np.zeros((0, T)))
        ud = torch.from_numpy(np.ascontiguousarray(np.stack(us, 0), dtype=np.float64)).to(dev)
        idx_d = torch.empty(R, K, dtype=torch.int32, device=dev)
        k.kpp_run_batch(self._permuted(Xc), self._row_norms(Xc), firsts, K, T, ud, idx_d)
        ih = idx_d.cpu().numpy().astype(np.int64)
        self.init_indices_log_ = [ih[r] for r in range(R)]          # every run's picks (the sequential loop appends run by run)
        self.init_indices_ = ih[-1]
        return [Xc.index_select(0, idx_d[r].long()).contiguous() for r in range(R)]

    def _kmeans_plusplus(self, Xc, K):
        """_kmeans_plusplus (_kmeans.py:174-277): RNG draws on the host from numpy's legacy RandomState (as
        sklearn), distances / potentials / cumsum-search on the device.  In sharded runs every rank seeds on the
        all-gathered matrix and follows rank 0's draws (identical centres everywhere)."""
        k = self.k
        if self._sharded:
            W = len(self._sizes)
            mx = int(self._sizes.max())
            if int(self._sizes.min()) == mx:
                Xs = self._gather(Xc).reshape(-1, Xc.shape[1]).contiguous()
            else:
                pad = torch.zeros(mx, Xc.shape[1], dtype=Xc.dtype, device=Xc.device)
                pad[: Xc.shape[0]] = Xc
                allp = self._gather(pad)
                Xs = torch.cat([allp[r, : int(self._sizes[r])] for r in range(W)], 0).contiguous()
        else:
            Xs = Xc
        N, Dp = Xs.shape
        dev = Xs.device
        rs = self._rng()
        T = 2 + int(np.log(K))
        newdist = torch.empty(T, N, dtype=torch.float32, device=dev)
        closest = torch.empty(N, dtype=torch.float32, device=dev)
        pot = torch.empty(T, dtype=torch.float64, device=dev)
        cand = torch.zeros(T, dtype=torch.int32, device=dev)
        vals = torch.empty(T, dtype=torch.float64, device=dev)
        idx = np.full(K, -1, np.int64)

        def bcast(a):
            if not self._sharded:
                return a
            t = torch.from_numpy(np.ascontiguousarray(a))
            t = t if torch.distributed.get_backend(self.process_group) == "gloo" else t.to(dev)
            src = torch.distributed.get_global_rank(self.process_group, 0)
            torch.distributed.broadcast(t, src, group=self.process_group)
            return t.cpu().numpy()

        first = int(bcast(np.array([rs.choice(N, p=np.full(N, 1.0 / N))], np.int64))[0])
        if hasattr(k, "kpp_run") and T <= 16:
            # the RNG draws of the loop below do not depend on the data: draw them all now (same stream, same order),
            # and let the device run the K - 1 steps back to back
            u = bcast(rs.uniform(size=(K - 1, T))) if K > 1 else np.zeros((0, T))
            ud = torch.from_numpy(np.ascontiguousarray(u, dtype=np.float64)).to(dev)
            idx_d = torch.empty(K, dtype=torch.int32, device=dev)
            Xp = xn = None
            if getattr(k, "uses_perm", False):
                Xp = self._permuted(Xs)                      # the same permuted copy the E-step uses (cached per fit)
                xn = self._row_norms(Xs)
            k.kpp_run(Xs, first, K, T, ud, idx_d, Xp, xn)
            self.init_indices_ = idx_d.cpu().numpy().astype(np.int64)
            self.init_indices_log_.append(self.init_indices_)
            return Xs.index_select(0, idx_d.long()).contiguous()
        idx[0] = first
        cand[0] = first
        k.kpp_step(Xs, cand, 1, None, closest, pot)
        cur_pot = float(pot[0].item())
        for c in range(1, K):
            rv = bcast(rs.uniform(size=T)) * cur_pot
            vals.copy_(torch.from_numpy(rv))
            k.cumsum_search(closest, vals, T, cand)
            k.kpp_step(Xs, cand, T, closest, newdist, pot)
            ph = pot.cpu().numpy()
            b = int(np.argmin(ph))
            cur_pot = float(ph[b])
            closest.copy_(newdist[b])
            idx[c] = int(cand[b].item())
        self.init_indices_ = idx
        self.init_indices_log_.append(idx)
        return Xs.index_select(0, torch.from_numpy(idx).to(dev)).contiguous()
