"""CPU, world_size 2 over gloo: BASELINE configs[2] as a pipeline (online_train.py:605-667) — eval loader with the
reference's shuffling + padding DistributedSampler -> extraction (resident shards or the reference's per-batch gather)
-> fit_cluster (row-sharded over the process group, k-means++ included, or on rank 0) -> labels in DATASET order on
every rank + vid_clusters.txt.  Driven with the oracle-backed kernel provider (tests/kmeans_cpu_kernels.py) and a
stand-in linear "encoder" — this covers the host logic; the same flow on the HIP kernels is tests/test_dist_gpu.py."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

N_DATA, K, FEAT = 37, 4, 16          # 37: not a multiple of the batch size or of the world size -> padding duplicates


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dataset():
    rng = np.random.default_rng(17)
    cen = rng.standard_normal((K, FEAT)) * 4
    y = rng.integers(0, K, N_DATA)
    x = (cen[y] + 0.2 * rng.standard_normal((N_DATA, FEAT))).astype(np.float32)
    return x, y


class _EvalSet(torch.utils.data.Dataset):
    def __init__(self):
        self.x, self.y = _dataset()

    def __len__(self):
        return len(self.x)

    def __getitem__(self, i):
        return torch.from_numpy(self.x[i]), int(self.y[i]), 0, i      # (clip, target, info, index): evaluate.py:156


def _encoder():
    torch.manual_seed(5)
    m = torch.nn.Linear(FEAT, 8)
    return m


def _cfg(out_dir, world, sharded, drop_last=False):
    ns = types.SimpleNamespace
    return ns(NUM_GPUS=world, OUTPUT_PATH=out_dir, DATASET=ns(POSITIVE_SAMPLING_P=0.2),
              ITERCLUSTER=ns(METHOD='kmeans', K=K, L2_NORMALIZE=True, FINCH_PARTITION=0, ADAPTIVEP=True, SHARDED=sharded))


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    torch.distributed.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    try:
        from video_similarity_search_amd.online_train import iterative_cluster_step, broadcast_cluster_labels
        from kmeans_cpu_kernels import OracleKernels
        ds = _EvalSet()
        enc = _encoder()
        res = {}
        for tag, sharded, drop_last in (("sharded", True, False), ("rank0", False, False), ("rank0_drop", False, True)):
            sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=3,
                                                                      drop_last=drop_last)
            loader = torch.utils.data.DataLoader(ds, batch_size=4, sampler=sampler, drop_last=drop_last)
            d = os.path.join(out_dir, tag)
            os.makedirs(d, exist_ok=True)
            cfg = _cfg(d, world, sharded)
            np.random.seed(1)
            labels, nmi = iterative_cluster_step(None, cfg, enc, loader, epoch=2, cuda=False, device="cpu",
                                                 is_master_proc=(rank == 0), kmeans_kernels=OracleKernels())
            res[tag] = labels
            res[tag + "_nmi"] = -1.0 if nmi is None else nmi
            res[tag + "_p"] = cfg.DATASET.POSITIVE_SAMPLING_P
            res[tag + "_seen"] = np.array(sorted(set(int(i) for b in loader for i in b[3])), np.int64)
        # a master-side failure must reach every rank as an exception, not as a hang inside dist.broadcast
        try:
            broadcast_cluster_labels(np.zeros(5, np.int32) if rank == 0 else None, N_DATA, "cpu", rank == 0)
            res["bad_raised"] = False
        except ValueError:
            res["bad_raised"] = True
        except Exception:
            res["bad_raised"] = False
        # sharded route: one rank failing its device / shard check must raise on EVERY rank, not strand the other in a collective
        class _Broken(OracleKernels):
            def check(self):
                if rank == 1:
                    raise RuntimeError("no device on this rank")
        sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=False)
        loader = torch.utils.data.DataLoader(ds, batch_size=4, sampler=sampler)
        d = os.path.join(out_dir, "broken")
        os.makedirs(d, exist_ok=True)
        try:
            iterative_cluster_step(None, _cfg(d, world, True), enc, loader, epoch=2, cuda=False, device="cpu",
                                   is_master_proc=(rank == 0), kmeans_kernels=_Broken())
            res["shard_fail_raised"] = False
        except RuntimeError as e:
            res["shard_fail_raised"] = "failed on a rank" in str(e)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), **res)
    finally:
        torch.distributed.destroy_process_group()


def test_extract_cluster_pipeline_two_ranks_gloo(tmp_path):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from kmeans_cpu_kernels import OracleKernels
    from sklearn.metrics import normalized_mutual_info_score as nmi_score
    from video_similarity_search_amd.clustering import fit_cluster
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (dict(np.load(os.path.join(tmp_path, f"r{r}.npz"))) for r in range(world))
    x, y = _dataset()
    with torch.no_grad():
        emb = _encoder()(torch.from_numpy(x))
    # single-process labels: the same clustering call over the whole (dataset-ordered) matrix
    np.random.seed(1)
    single = fit_cluster(emb, 'kmeans', K, True, kernels=OracleKernels())
    for tag in ("sharded", "rank0"):
        a, b = r0[tag], r1[tag]
        assert a.dtype == np.int32 and a.shape == (N_DATA,)
        assert np.array_equal(a, b), tag                                   # every rank holds the same dataset-ordered labels
        assert a.min() >= 0 and a.max() < K                                # padding never leaves a slot empty
        assert nmi_score(a, single) == pytest.approx(1.0), tag             # same partition as the single-process run
        lines = open(os.path.join(tmp_path, tag, "vid_clusters.txt")).read().split()
        assert [int(v) for v in lines] == a.tolist()                       # the file IS the dataset-ordered array
        assert r0[tag + "_nmi"] == pytest.approx(nmi_score(y, a), abs=0.05)
        assert float(r0[tag + "_p"]) == pytest.approx(1.0 - float(r0[tag + "_nmi"]))      # ADAPTIVEP on the master (:644-645)
    # sharded and rank-0 routes agree on the partition (labels may be permuted: different k-means++ row order)
    assert nmi_score(r0["sharded"], r0["rank0"]) == pytest.approx(1.0)
    # drop_last: the loader skips rows -> their slots are -1 (the reference writes 'None'), everything seen is labelled
    a = r0["rank0_drop"]
    assert np.array_equal(a, r1["rank0_drop"])
    seen = np.union1d(r0["rank0_drop_seen"], r1["rank0_drop_seen"])
    assert len(seen) < N_DATA
    assert (a[seen] >= 0).all() and (np.delete(a, seen) == -1).all()
    lines = open(os.path.join(tmp_path, "rank0_drop", "vid_clusters.txt")).read().split()
    assert lines.count("None") == N_DATA - len(seen)
    assert bool(r0["bad_raised"]) and bool(r1["bad_raised"]) is False      # rank 1 gets the -2 marker, not an exception
    assert bool(r0["shard_fail_raised"]) and bool(r1["shard_fail_raised"])


def test_dataset_order_helper():
    from video_similarity_search_amd.online_train import _dataset_order
    o = _dataset_order(6, [4, 0, 2, 0], [7, 1, 3, 9])
    assert o.tolist() == [9, -1, 3, -1, 7, -1] and o.dtype == np.int32     # last duplicate wins, unseen slots -1
    with pytest.raises(ValueError):
        _dataset_order(3, [0, 5], [1, 1])
    with pytest.raises(ValueError):
        _dataset_order(3, [0, 1], [1])
