"""GPU: the multi-GPU paths on RCCL process groups — DistributedDataParallel around the
encoder, the row-sharded HIP k-means (slic_kmeans_combine_shards), the extract -> sharded fit_cluster -> labels
pipeline of configs[2], misc.distributed_helper.launch_processes, and bench.py's own launcher.

Each case runs tests/dist_gpu_worker.py under `python -m torch.distributed.run` in child processes (the parent pytest
process never initialises a process group).  W = 1 runs everywhere (a one-rank RCCL group takes the W > 1 code paths);
the W = 2 variants skip themselves on a box with fewer than two GPUs."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(case, world, out_dir, timeout=900, extra_env=None):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "dist_gpu_worker.py"), case, str(out_dir)]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-4000:]
    return [dict(np.load(os.path.join(out_dir, f"{case}_r{r}.npz"))) for r in range(world)]


def _worlds():
    return [pytest.param(1, id="w1"),
            pytest.param(2, id="w2", marks=pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs"))]


@pytest.mark.parametrize("world", _worlds())
def test_ddp_step_equals_plain_step(gpu, tmp_path, world):
    """online_train.py:485-494: DDP(model) gradients == mean over ranks of the un-wrapped gradients, bit for bit
    (six autograd segments, DDP's bucketed RCCL all-reduce); replicas stay identical after the optimiser step"""
    res = _run("ddp", world, tmp_path)
    for r in res:
        assert int(r["n_params"]) == 66                                              # every parameter tensor of the R3D-18
        assert int(r["n_not_bit_equal"]) == 0, float(r["worst_abs"])
        assert float(r["l_plain"]) == float(r["l_ddp"])
        assert bool(r["replicas_equal"])
    for k in res[0]:
        if k.startswith("g/"):
            assert all(np.array_equal(res[0][k], r[k]) for r in res[1:]), k            # all ranks hold the same reduced gradient


@pytest.mark.parametrize("world", _worlds())
def test_data_parallel_fast_wrapper_equals_plain_steps(gpu, tmp_path, world):
    """misc.distributed_helper.data_parallel (round 6: DistributedDataParallel with its per-step copies removed — flat buffer broadcast,
    gradients written into the bucket views, ReduceOp.AVG hook): four training steps bit-equal to the un-wrapped model with an explicit
    gradient mean, state_dict keys untouched, and from the third step on EVERY gradient already aliases its bucket when the reducer sees it"""
    res = _run("ddp_fast", world, tmp_path)
    for r in res:
        assert int(r["n_params"]) == 66
        assert int(r["n_grad_not_equal"]) == 0 and int(r["n_weight_not_equal"]) == 0
        assert bool(r["bufs_equal_ref"]) and bool(r["keys_same"]) and int(r["nbt"]) == 4
        al = [int(v) for v in r["aliased"]]
        assert al[0] == 0 and al[2] == 66 and al[3] == 66, al          # step 1: nothing handed over yet; step 2: views of the first buckets (rebuilt since)


@pytest.mark.parametrize("world", _worlds())
def test_sync_batchnorm_equals_full_batch(gpu, tmp_path, world):
    """cfg.SYNC_BATCH_NORM (online_train.py:466-468): SyncBatchNorm over W ranks with B / W clips each == plain BatchNorm over
    the B clips in one process — embeddings, parameter gradients summed over the ranks, running statistics (fp32 tolerance:
    the statistics are merged in a different association) — and == the CPU oracle's plain BatchNorm on the same four clips"""
    res = _run("syncbn", world, tmp_path)
    for r in res:
        assert int(r["n_sync"]) == 21
        assert float(r["emb_err"]) < 1e-5, float(r["emb_err"])
        assert float(r["worst_grad_rel"]) < 5e-4, float(r["worst_grad_rel"])
        assert float(r["worst_running_rel"]) < 1e-5, float(r["worst_running_rel"])
    # PARITY, not only the property: the same four clips through the CPU oracle's plain BatchNorm (fp32 and fp64) — embeddings within
    # north_star's 1e-4, the ranks' summed gradients of a deep and a shallow tensor within 1e-3 of fp64 (relative to the largest entry)
    from oracle import encoder as oe
    sd0 = {k[4:]: v for k, v in res[0].items() if k.startswith("sd0/")}
    rng = np.random.default_rng(77)
    xfull = torch.from_numpy(rng.standard_normal((4, 3, 8, 32, 32)).astype(np.float32))
    wsum = torch.from_numpy(rng.standard_normal((4, 32)).astype(np.float32))
    t64 = oe.to_torch(sd0, dtype=torch.float64, requires_grad=True)
    e64 = oe.encoder_forward(t64, xfull.double(), training=True)
    names = ["conv1.weight", "layer4.1.conv2.weight"]
    g64 = torch.autograd.grad((e64 * wsum.double()).sum(), [t64[k] for k in names])
    with torch.no_grad():
        e32 = oe.encoder_forward(oe.to_torch(sd0), xfull, training=True)
    per = 4 // world
    for rk, r in enumerate(res):
        emb = torch.from_numpy(r["emb"])
        assert (emb - e32[rk * per:(rk + 1) * per]).abs().max().item() <= 1e-4
        assert (emb.double() - e64.detach()[rk * per:(rk + 1) * per]).abs().max().item() <= 1e-4
    for k, ref in zip(names, g64):
        d = (torch.from_numpy(res[0]["grad/" + k]).double() - ref).abs().max().item() / ref.abs().max().item()
        assert d <= 1e-3, (k, d)


@pytest.mark.parametrize("world", _worlds())
def test_sharded_hip_kmeans_equals_oracle(gpu, golden_dir, tmp_path, world):
    """KMeans(process_group=WORLD) on the HIP kernels == the oracle with n_shards = -W (fp64 all-reduce exchange) / W (all-gather
    + rank-ordered add) == sklearn's golden labels.  k-means++ in a sharded run: same rows on every rank, and the
    run from those rows equals the oracle's from the same rows."""
    from oracle import kmeans as ok
    res = _run("kmeans", world, tmp_path)
    r0 = res[0]
    for name, ex in [(n, e) for e in ("allreduce", "allgather") for n in ("clustered_empty", "d128", "unstructured")]:
        g = dict(np.load(os.path.join(golden_dir, f"kmeans_{name}.npz")))
        X, init = g["X"], g["init"]
        mean = ok.col_mean(X)
        Xc = X - mean
        ref = ok.lloyd(Xc, init - mean, tol_abs=ok.tolerance(Xc, 1e-4), n_shards=-world if ex == "allreduce" else world, trace=True)
        bare, name = name, f"{ex}/{name}"
        assert np.array_equal(r0[f"{name}/labels"], ref["labels"]), name
        assert np.array_equal(r0[f"{name}/labels"], g["labels"]), name
        assert int(r0[f"{name}/n_iter"]) == ref["n_iter"] == int(g["n_iter"])
        assert bool(r0[f"{name}/strict"]) == ref["strict"]
        np.testing.assert_array_equal(r0[f"{name}/centers"], (ref["centers"] + mean).astype(np.float32))
        assert float(r0[f"{name}/inertia"]) == pytest.approx(ref["inertia"], rel=1e-9)
        per = (len(X) + world - 1) // world
        for rk, r in enumerate(res):
            assert np.array_equal(r[f"{name}/centers"], r0[f"{name}/centers"])          # bit-identical centres on all ranks
            tr = r[f"{name}/trace_local"]
            for it in range(ref["n_iter"]):
                assert np.array_equal(tr[it], ref["trace"][it][rk * per:(rk + 1) * per]), (name, rk, it)
        if bare == "clustered_empty":
            assert int(r0[f"{name}/nreloc"]) >= 1
    # k-means++ branch
    g = dict(np.load(os.path.join(golden_dir, "kmeans_d128.npz")))
    X = g["X"]
    idx = r0["kpp/init_indices"]
    assert len(set(idx.tolist())) == 12 and idx.min() >= 0 and idx.max() < len(X)
    for r in res[1:]:
        assert np.array_equal(r["kpp/init_indices"], idx) and np.array_equal(r["kpp/centers"], r0["kpp/centers"])
    mean = ok.col_mean(X)
    Xc = X - mean
    ref = ok.lloyd(Xc, Xc[idx], tol_abs=ok.tolerance(Xc, 1e-4), n_shards=-world)
    labels = np.concatenate([r["kpp/labels_local"] for r in res])
    assert np.array_equal(labels, ref["labels"]) and int(r0["kpp/n_iter"]) == ref["n_iter"]
    assert float(r0["kpp/inertia"]) == pytest.approx(ref["inertia"], rel=1e-9)


@pytest.mark.parametrize("world", _worlds())
def test_extract_sharded_cluster_pipeline(gpu, tmp_path, world):
    """configs[2] end to end on the GPU: eval-mode encoder with resident shards -> sharded fit_cluster -> dataset-ordered
    labels on every rank + vid_clusters.txt; same partition as clustering the gathered embeddings in one process"""
    from sklearn.metrics import normalized_mutual_info_score as nmi
    from video_similarity_search_amd.clustering import fit_cluster
    res = _run("pipeline", world, tmp_path)
    r0 = res[0]
    n = int(r0["n"])
    lab = r0["labels"]
    assert lab.shape == (n,) and lab.dtype == np.int32 and lab.min() >= 0 and lab.max() < 4
    for r in res[1:]:
        assert np.array_equal(r["labels"], lab)
    lines = open(os.path.join(tmp_path, "vid_clusters.txt")).read().split()
    assert [int(v) for v in lines] == lab.tolist()
    # the reference-shaped route on the same embeddings: gathered matrix, one process
    emb, idxs = r0["emb"], r0["idxs"]
    assert emb.shape[0] >= n and set(idxs.tolist()) == set(range(n))
    np.random.seed(1)
    single = fit_cluster(torch.from_numpy(emb), 'kmeans', 4, True)
    order = np.full(n, -1, np.int64)
    order[idxs] = single
    assert nmi(order, lab) > 0.99


@pytest.mark.parametrize("world", _worlds())
def test_configs3_full_size_ddp_step(gpu, tmp_path, world):
    """BASELINE configs[3] per-GPU step (39 clips of 3x16x112x112 through one forward, random_semi_hard + LLC, DDP over RCCL):
    two steps of triplet_train_epoch on the real R3D-18 — finite, weights moved, replicas identical, loss averaged over ranks"""
    res = _run("step39", world, tmp_path, timeout=1500)
    for r in res:
        assert bool(r["finite"]) and float(r["moved"]) > 0 and bool(r["replicas_equal"]) and int(r["nbt"]) == 2
        assert np.isfinite(float(r["avg"])) and float(r["avg"]) > 0
    assert all(float(r["avg"]) == float(res[0]["avg"]) for r in res)        # the logged loss is the all-reduced mean


def _launched(cmd_args, cfg):
    """func of launch_processes (online_train.py:787): runs in every spawned rank"""
    import torch.distributed as dist
    from video_similarity_search_amd.misc import distributed_helper as du
    t = torch.ones(3, device="cuda") * (dist.get_rank() + 1)
    du.all_reduce([t], avg=False)
    g = du.all_gather([torch.arange(2, device="cuda") + 10 * dist.get_rank()])[0]
    W = dist.get_world_size()
    assert du.get_world_size() == W == cfg.NUM_GPUS and t[0].item() == W * (W + 1) / 2 and g.numel() == 2 * W
    if du.is_master_proc():
        with open(cmd_args, "w") as f:
            f.write(f"ok {W}")
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="launch_processes spawns only for NUM_GPUS > 1")
def test_launch_processes_two_gpus(gpu, tmp_path):
    """misc/distributed_helper.py:30-37: mp.spawn of one process per GPU + RCCL init, from a parent that has not
    initialised the GPU (run in a child interpreter for that reason)"""
    import types
    marker = os.path.join(tmp_path, "ok.txt")
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {HERE!r})\n"
            "import types\n"
            "from video_similarity_search_amd.misc import distributed_helper as du\n"
            "import test_dist_gpu as t\n"
            f"du.launch_processes({marker!r}, types.SimpleNamespace(NUM_GPUS=2), t._launched, 0, 1, 'tcp://127.0.0.1:{_free_port()}')\n")
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    assert open(marker).read() == "ok 2"


@pytest.mark.parametrize("world", [2, 3])
def test_oneshot_exchange_two_processes_one_gpu(gpu, golden_dir, tmp_path, world):
    """exchange='oneshot' (slic_allreduce_oneshot_f64, csrc/oneshot.hip: IPC-mapped inboxes, one kernel per exchange) with TWO (and THREE)
    processes on the ONE leased GPU (IPC handles work on the same device; RCCL does not): 25 raw exchanges exact, and the sharded HIP
    k-means == the oracle with n_shards = -W (fp64 combine), every iteration's labels bit-equal, == sklearn's golden labels"""
    from oracle import kmeans as ok
    # (a bounded wait of 20 s per exchange instead of the default five minutes: a broken exchange fails this test in seconds)
    res = _run("oneshot", world, tmp_path, timeout=600, extra_env={"SLIC_TEST_SAME_GPU": "1", "SLIC_COMM_TIMEOUT_MS": "20000"})
    for rk, r in enumerate(res):
        assert bool(r["raw_ok"]) and int(r["n_exchanges"]) == 25
        assert bool(r["stress_ok"])                                   # 300 exchanges with no host synchronisation in between
        assert [int(v) for v in r["info"][:2]] == [world, rk] and int(r["info"][3]) >= 325       # world, rank, exchanges issued
    for name in ("clustered_empty", "d128", "unstructured"):
        g = dict(np.load(os.path.join(golden_dir, f"kmeans_{name}.npz")))
        X, init = g["X"], g["init"]
        mean = ok.col_mean(X)
        Xc = X - mean
        ref = ok.lloyd(Xc, init - mean, tol_abs=ok.tolerance(Xc, 1e-4), n_shards=-world, trace=True)
        labels = np.concatenate([r[f"{name}/labels_local"] for r in res])
        trace = np.concatenate([r[f"{name}/trace_local"] for r in res], axis=1)
        assert "oneshot" in str(res[0][f"{name}/comm"])
        assert int(res[0][f"{name}/n_iter"]) == ref["n_iter"] == int(g["n_iter"])
        assert np.array_equal(labels, ref["labels"]) and np.array_equal(labels, g["labels"])
        assert np.array_equal(trace, np.asarray(ref["trace"]))
        assert all(np.array_equal(res[0][f"{name}/centers"], r[f"{name}/centers"]) for r in res[1:])      # replicas bit-identical
        np.testing.assert_array_equal(res[0][f"{name}/centers"], (ref["centers"] + mean).astype(np.float32))


def test_oneshot_lost_peer_times_out(gpu, tmp_path):
    """a peer that never pushes: the exchange kernel's wait is bounded (SLIC_COMM_TIMEOUT_MS), the stream completes, slic_oneshot_check
    raises SLIC_ETIMEOUT and the communicator refuses further exchanges — seconds, not a hang"""
    res = _run("oneshot_timeout", 2, tmp_path, timeout=300, extra_env={"SLIC_TEST_SAME_GPU": "1", "SLIC_COMM_TIMEOUT_MS": "2000"})
    r0 = res[0]
    assert bool(r0["raised"]) and "did not receive every peer" in str(r0["msg"]) and bool(r0["refused_after"])
    assert 1.5 < float(r0["seconds"]) < 30.0


def _bench(args, timeout=1500):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_self_launch_one_rank(gpu):
    """`python bench.py --gpus N` typed plainly starts its own torch.distributed.run child (before any GPU call in the
    parent) and relays rank 0's JSON line: exercised with N = 1 (+ --force-dist: DDP and the sharded k-means over a
    one-rank RCCL group), the same code path the driver takes for N = 2, 4, 8"""
    r = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--self-launch", "--force-dist", "--no-cpu-baseline", "--quick"])
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["value"] > 0 and r["config"]["parallelism"] == "dp1"
    assert "sharded" in r["secondary"]["config"]["workload"]
    # the fields that make a multi-GPU run self-verifying (what RCCL really reduced over, not what WORLD_SIZE claims)
    d = r["distributed"]
    assert d["initialised"] and d["backend"] == "nccl" and d["rccl_ranks_seen"] == r["n_gpus"] == d["world_size_env"]
    assert d["rccl_version"].count(".") == 2 and len(d["ms_per_step_per_rank"]) == 1
    assert d["ms_per_step_min"] <= d["ms_per_step_max"] <= r["ms_per_step"] * 1.001 + 1e-6
    assert d["ddp"]["buckets"] >= 1 and d["ddp"]["gradient_as_bucket_view"] in (True, False) and d["ddp"]["gradient_bytes"] > 100e6
    ex = r["secondary"]["exchange"]
    assert ex["kind"] in ("allreduce", "allgather", "oneshot") and ex["collectives_per_iteration"] == 1 and ex["payload_bytes_per_rank"] > 0
    assert "torch.distributed" in ex["communicator"] or "slic" in ex["communicator"]
    assert "weak_scaled" in r["secondary"] and r["secondary"]["weak_scaled"]["n_gpus"] == 1


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_bench_two_gpus_plain_invocation(gpu):
    r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--quick"])
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 64 and r["scaling"] == "weak"
    assert r["secondary"]["n_gpus"] == 2 and r["distributed"]["rccl_ranks_seen"] == 2 and len(r["distributed"]["ms_per_step_per_rank"]) == 2
