"""Worker of tests/test_dist_gpu.py: one rank of an RCCL (backend "nccl") process group on real MI355X(s).
Started by `python -m torch.distributed.run --nproc-per-node W tests/dist_gpu_worker.py <case> <out_dir>`; a
one-rank group (W = 1) takes exactly the code paths of W > 1 (DDP reducer, all-gather of the centroid partials,
rank-ordered combine, label all-gather), which is how a single-GPU box covers them.  Every case writes
<out_dir>/<case>_r<rank>.npz; the parent test compares the ranks with each other and with the oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

TINY = dict(hidden_layer=64, out_dim=32, num_classes=101, n_input_channels=3, shortcut_type='B', conv1_t_size=7,
            conv1_t_stride=1, no_max_pool=True, widen_factor=0.125, projection_head=True, predict_temporal_ds=False,
            spatio_temporal_attention=False, classifier=False, dropout=None)


def tiny_state_dict(seed=3):
    """seeded tiny R3D-18 weights with the reference's init rules — built without the oracle so the worker is product-only"""
    import contextlib
    import io
    from video_similarity_search_amd.models import generate_model
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        m = generate_model(18, **TINY)
    return m, {k: v.clone() for k, v in m.state_dict().items()}


def case_ddp(rank, world, out):
    """DistributedDataParallel(model) step == the un-wrapped step followed by an explicit gradient mean over the ranks
    (online_train.py:485-494): bit-equal for W = 1 and W = 2 (a two-term sum is order-free, /2 is exact)."""
    from video_similarity_search_amd.loss import OnlineTripletLoss
    m, sd0 = tiny_state_dict()
    m = m.cuda().train()
    rng = np.random.default_rng(100 + rank)                      # every rank its own clips
    x = torch.from_numpy(rng.standard_normal((4, 3, 8, 32, 32)).astype(np.float32)).cuda()
    labels = torch.arange(2).repeat(2).cuda()
    crit = OnlineTripletLoss(0.2, 'cosine')

    def run(model):
        model.zero_grad(set_to_none=True)
        loss, _ = crit(model(x), labels, sampling_strategy='noise_contrastive')
        loss.backward()
        return float(loss.item())

    l_plain = run(m)
    plain = {k: p.grad.clone() for k, p in m.named_parameters()}
    for g in plain.values():                                       # what DDP is specified to produce
        dist.all_reduce(g)
        g.div_(world)
    m.load_state_dict(sd0)                                         # BN running stats back to the start
    ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[torch.cuda.current_device()])
    l_ddp = run(ddp)
    res = dict(l_plain=l_plain, l_ddp=l_ddp)
    nbad = 0
    worst = 0.0
    for k, p in m.named_parameters():
        same = torch.equal(p.grad, plain[k])
        nbad += int(not same)
        worst = max(worst, float((p.grad - plain[k]).abs().max().item()))
        res["g/" + k] = p.grad.cpu().numpy()
    res.update(n_params=len(plain), n_not_bit_equal=nbad, worst_abs=worst)
    # one optimiser step under DDP keeps the replicas identical
    opt = torch.optim.SGD(ddp.parameters(), lr=0.1, momentum=0.5)
    opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    allw = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(allw, flat)
    res["replicas_equal"] = all(torch.equal(allw[0], w) for w in allw)
    np.savez(out, **res)


def case_ddp_fast(rank, world, out):
    """misc.distributed_helper.data_parallel — DistributedDataParallel without its per-step copies (flat buffer broadcast, gradients written
    into the bucket views, ReduceOp.AVG hook) — over FOUR steps (the reducer rebuilds its buckets after the first; from the third on the
    engine writes into the bucket views it was handed): every step's gradients and the weights after every SGD step bit-equal to the
    un-wrapped model trained beside it with an explicit gradient mean over the ranks; running statistics equal to rank 0's on every rank;
    state_dict keys untouched; from step 3 on every gradient aliases a bucket (no copy left for the reducer)"""
    from video_similarity_search_amd.loss import OnlineTripletLoss
    from video_similarity_search_amd.misc.distributed_helper import data_parallel
    from video_similarity_search_amd.models import resnet as rn
    m, sd0 = tiny_state_dict()
    m = m.cuda().train()
    ref, _ = tiny_state_dict()
    ref.load_state_dict(sd0)
    ref = ref.cuda().train()
    keys0 = list(m.state_dict().keys())
    crit = OnlineTripletLoss(0.2, 'cosine')
    labels = torch.arange(2).repeat(2).cuda()
    ddp = data_parallel(m, torch.cuda.current_device())
    assert list(m.state_dict().keys()) == keys0 and list(ddp.state_dict().keys()) == ["module." + k for k in keys0]
    info = ddp.slic_ddp
    assert info["flat_buffers"] and info["buffer_broadcasts_per_forward"] == 2 and info["buffers_flattened"] == 63
    bn = [b for b in m.buffers() if b.dtype == torch.float32]
    assert len({b.untyped_storage().data_ptr() for b in bn}) == 1               # one flat tensor behind the 42 running statistics
    opt_d = torch.optim.SGD(ddp.parameters(), lr=0.05, momentum=0.5)
    opt_r = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.5)
    nbad = nbad_w = 0
    aliased = []
    for step in range(4):
        rng = np.random.default_rng(1000 * step + rank)                 # every rank its own clips, new ones every step
        x = torch.from_numpy(rng.standard_normal((4, 3, 8, 32, 32)).astype(np.float32)).cuda()
        # reference: plain model, explicit mean of the gradients; its buffers follow rank 0 like DistributedDataParallel's
        for b in ref.buffers():
            dist.broadcast(b, 0)
        loss_r, _ = crit(ref(x), labels, sampling_strategy='noise_contrastive')
        opt_r.zero_grad()
        loss_r.backward()
        for p_ in ref.parameters():
            dist.all_reduce(p_.grad)
            p_.grad.div_(world)
        loss_d, _ = crit(ddp(x), labels, sampling_strategy='noise_contrastive')
        opt_d.zero_grad()                                               # the reference's order: after the forward (online_train.py:101-103)
        views = dict(getattr(m, "_slic_grad_views", {}))
        loss_d.backward()
        assert float(loss_d.item()) == float(loss_r.item())
        for (k, p_), q_ in zip(m.named_parameters(), ref.parameters()):
            nbad += int(not torch.equal(p_.grad, q_.grad))
        aliased.append(sum(int(p_ in views and p_.grad.untyped_storage().data_ptr() == views[p_].untyped_storage().data_ptr()) for p_ in m.parameters()))
        opt_r.step()
        opt_d.step()
        for p_, q_ in zip(m.parameters(), ref.parameters()):
            nbad_w += int(not torch.equal(p_, q_))
    bufs_equal = all(torch.equal(a, b) for a, b in zip(m.buffers(), ref.buffers()))
    flat = torch.cat([b.detach().reshape(-1).double() for b in m.buffers()])
    allb = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(allb, flat)
    # (after the last forward every rank updated its own statistics: compare what the broadcast delivered, i.e. the counters)
    sd_after = ddp.module.state_dict()
    np.savez(out, n_grad_not_equal=nbad, n_weight_not_equal=nbad_w, aliased=np.array(aliased), n_params=len(list(m.parameters())),
             bufs_equal_ref=bufs_equal, keys_same=(list(sd_after.keys()) == keys0),
             nbt=int(sd_after["bn1.num_batches_tracked"].item()))


def case_kmeans(rank, world, out):
    """KMeans(process_group=WORLD) on the HIP kernels: rows sharded, slic_kmeans_lloyd_local -> ONE collective over RCCL (fp64
    all-reduce of [sums | counts | n_changed], or all-gather of the fp32 payloads) -> slic_kmeans_lloyd_global; explicit init
    (goldens, both exchanges) and the k-means++ branch"""
    from video_similarity_search_amd.clustering.kmeans_hip import KMeans
    res = {}
    for name, ex in [(n, e) for e in ("allreduce", "allgather") for n in ("clustered_empty", "d128", "unstructured")]:
        g = dict(np.load(os.path.join(HERE, "golden", f"kmeans_{name}.npz")))
        X, init = g["X"], g["init"]
        N = len(X)
        per = (N + world - 1) // world
        shard = torch.from_numpy(X[rank * per:(rank + 1) * per]).cuda()
        km = KMeans(n_clusters=init.shape[0], init=init, n_init=1, process_group=dist.group.WORLD, trace=True, exchange=ex).fit(shard)
        name = f"{ex}/{name}"
        lab = torch.from_numpy(km.labels_).cuda()
        sizes = [min(per, N - r * per) for r in range(world)]
        parts = [torch.empty(s, dtype=lab.dtype, device="cuda") for s in sizes]
        dist.all_gather(parts, lab) if len(set(sizes)) == 1 else _all_gather_ragged(parts, lab, sizes)
        res[f"{name}/labels"] = torch.cat(parts).cpu().numpy()
        res[f"{name}/centers"] = km.cluster_centers_
        res[f"{name}/n_iter"] = km.n_iter_
        res[f"{name}/strict"] = km.strict_
        res[f"{name}/inertia"] = km.inertia_
        res[f"{name}/nreloc"] = km.n_relocations_
        res[f"{name}/trace_local"] = km.trace_
    # k-means++ (sklearn _kmeans.py:174-277) in a sharded run: all ranks must pick the same rows of the GLOBAL matrix
    g = dict(np.load(os.path.join(HERE, "golden", "kmeans_d128.npz")))
    X = g["X"]
    N = len(X)
    per = (N + world - 1) // world
    shard = torch.from_numpy(X[rank * per:(rank + 1) * per]).cuda()
    km = KMeans(n_clusters=12, n_init=2, random_state=5 + rank * 0, process_group=dist.group.WORLD).fit(shard)
    res["kpp/init_indices"] = np.asarray(km.init_indices_)
    res["kpp/labels_local"] = km.labels_
    res["kpp/centers"] = km.cluster_centers_
    res["kpp/inertia"] = km.inertia_
    res["kpp/n_iter"] = km.n_iter_
    np.savez(out, **res)


def _all_gather_ragged(parts, t, sizes):
    mx = max(sizes)
    pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
    pad[: t.numel()] = t
    buf = [torch.empty_like(pad) for _ in sizes]
    dist.all_gather(buf, pad)
    for p, b, s in zip(parts, buf, sizes):
        p.copy_(b[:s])


class _EvalSet(torch.utils.data.Dataset):
    """synthetic eval-mode dataset with the reference loader's item shape: (clip, target, info, index)"""

    def __init__(self, n, seed=17):
        rng = np.random.default_rng(seed)
        self.x = rng.standard_normal((n, 3, 8, 32, 32)).astype(np.float32)
        self.y = rng.integers(0, 5, n)

    def __len__(self):
        return len(self.x)

    def __getitem__(self, i):
        return torch.from_numpy(self.x[i]), int(self.y[i]), 0, i


def case_pipeline(rank, world, out):
    """BASELINE configs[2] as a pipeline (online_train.py:605-667): extract with each rank's shard resident on its GPU ->
    sharded fit_cluster -> labels handed to every rank in dataset order -> vid_clusters.txt on rank 0.
    The DistributedSampler shuffles and pads exactly like the reference's eval loader (datasets/data_loader.py:283)."""
    import types
    from video_similarity_search_amd.online_train import iterative_cluster_step
    m, _ = tiny_state_dict(seed=11)
    m = m.cuda()
    n = 37                                                          # not a multiple of the batch or the world size
    ds = _EvalSet(n)
    sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=3)
    loader = torch.utils.data.DataLoader(ds, batch_size=4, sampler=sampler, drop_last=False)
    ns = types.SimpleNamespace
    out_dir = os.path.dirname(out)
    cfg = ns(NUM_GPUS=max(world, 2), OUTPUT_PATH=out_dir, DATASET=ns(POSITIVE_SAMPLING_P=0.2),
             ITERCLUSTER=ns(METHOD='kmeans', K=4, L2_NORMALIZE=True, FINCH_PARTITION=0, ADAPTIVEP=False, SHARDED=True))
    np.random.seed(1)
    labels, nmi = iterative_cluster_step(None, cfg, m, loader, epoch=0, cuda=True, device="cuda", is_master_proc=(rank == 0))
    res = dict(labels=np.asarray(labels), n=n)
    # the same embeddings through the reference-shaped (gathered, rank-0) path for comparison
    from video_similarity_search_amd.evaluate import evaluate
    emb, tg, idxs = evaluate(m, loader, gather=True)
    res.update(emb=emb.numpy(), idxs=np.asarray(idxs), targets=np.asarray(tg))
    np.savez(out, **res)


def case_step39(rank, world, out):
    """BASELINE configs[3]'s per-GPU step at full size: the real R3D-18 under DistributedDataParallel, 13 anchors + 13 positives
    + 13 second anchors of 3 x 16 x 112 x 112 through ONE forward, random_semi_hard mining + the LLC margin term, SGD —
    driven by triplet_train_epoch (online_train.py:255-392) for two steps"""
    import contextlib
    import io
    import types
    from video_similarity_search_amd.loss import OnlineTripletLoss
    from video_similarity_search_amd.models import generate_model
    from video_similarity_search_amd.online_train import triplet_train_epoch
    kw = dict(TINY, widen_factor=1.0, hidden_layer=2048, out_dim=128)
    torch.manual_seed(7)
    with contextlib.redirect_stdout(io.StringIO()):
        m = generate_model(18, **kw).cuda()
    ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[torch.cuda.current_device()])
    opt = torch.optim.SGD(ddp.parameters(), lr=0.1, momentum=0.5)
    rng = np.random.default_rng(50 + rank)
    b = 13
    batches = []
    for _ in range(2):
        views = [torch.from_numpy(rng.standard_normal((b, 3, 16, 112, 112)).astype(np.float32)) for _ in range(3)]
        batches.append((views, (torch.arange(b), torch.arange(b)), torch.arange(b)))

    class Loader(list):
        dataset = list(range(2 * b * world))

    ns = types.SimpleNamespace
    cfg = ns(LOSS=ns(LOCAL_LOCAL_CONTRAST=True, RELATIVE_SPEED_PERCEPTION=False, INTRA_NEGATIVE=False, DIST_METRIC='cosine',
                     LOCAL_LOCAL_MARGIN=0.04, LOCAL_LOCAL_WEIGHT=1.0),
             DATASET=ns(SAMPLING_STRATEGY='random_semi_hard'), NUM_GPUS=max(world, 2), TRAIN=ns(LOG_INTERVAL=1000),
             OUTPUT_PATH=os.path.dirname(out))
    w0 = m.conv1.weight.detach().clone()
    with contextlib.redirect_stdout(io.StringIO()):
        avg = triplet_train_epoch(Loader(batches), ddp, OnlineTripletLoss(0.2, 'cosine'), opt, 0, cfg, True, "cuda", rank == 0)
    flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    allw = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(allw, flat)
    np.savez(out, avg=avg, finite=bool(torch.isfinite(flat).all()), moved=float((m.conv1.weight.detach() - w0).abs().max()),
             replicas_equal=all(torch.equal(allw[0], w) for w in allw), nbt=int(m.bn1.num_batches_tracked))


def case_syncbn(rank, world, out):
    """cfg.SYNC_BATCH_NORM (online_train.py:466-468): convert_sync_batchnorm(model) trained on W ranks with B / W clips each ==
    the plain-BatchNorm model on all B clips in one process: same embeddings for this rank's clips, parameter gradients
    summed over the ranks == the full-batch gradients, same running statistics.  W = 1 takes the same collective path."""
    import copy
    B = 4
    m, sd0 = tiny_state_dict()
    m = m.cuda().train()
    rng = np.random.default_rng(77)                               # the SAME full batch on every rank
    xfull = torch.from_numpy(rng.standard_normal((B, 3, 8, 32, 32)).astype(np.float32)).cuda()
    wsum = torch.from_numpy(rng.standard_normal((B, 32)).astype(np.float32)).cuda()      # a fixed linear functional as the loss
    # reference: plain BatchNorm over the whole batch
    m.zero_grad(set_to_none=True)
    emb_full = m(xfull)
    (emb_full * wsum).sum().backward()
    gfull = {k: p.grad.clone() for k, p in m.named_parameters()}
    stats_full = {k: v.clone() for k, v in m.state_dict().items() if "running" in k}
    # SyncBatchNorm on this rank's slice
    ms = copy.deepcopy(m)
    ms.load_state_dict(sd0)
    ms = torch.nn.SyncBatchNorm.convert_sync_batchnorm(ms).cuda().train()
    lo, hi = rank * (B // world), (rank + 1) * (B // world)
    ms.zero_grad(set_to_none=True)
    emb = ms(xfull[lo:hi])
    (emb * wsum[lo:hi]).sum().backward()
    res = dict(emb_err=float((emb - emb_full[lo:hi]).abs().max() / emb_full.abs().max()))
    worst_g, worst_s = 0.0, 0.0
    for k, p in ms.named_parameters():
        g = p.grad.clone()
        dist.all_reduce(g)                                        # sum over ranks of the rank-local gradients
        worst_g = max(worst_g, float((g - gfull[k]).abs().max() / gfull[k].abs().max().clamp_min(1e-12)))
    for k, v in ms.state_dict().items():
        if "running" in k:
            worst_s = max(worst_s, float((v - stats_full[k]).abs().max() / stats_full[k].abs().max().clamp_min(1e-12)))
    res.update(worst_grad_rel=worst_g, worst_running_rel=worst_s, n_sync=sum(isinstance(x, torch.nn.SyncBatchNorm) for x in ms.modules()))
    # for the parent's comparison with the CPU oracle (the worker itself stays product-only): this rank's embeddings, the full-batch
    # gradient of one deep and one shallow tensor as the ranks' sum, and — from rank 0 — the initial weights
    res["emb"] = emb.detach().cpu().numpy()
    for k in ("conv1.weight", "layer4.1.conv2.weight"):
        g = dict(ms.named_parameters())[k].grad.clone()
        dist.all_reduce(g)
        res["grad/" + k] = g.cpu().numpy()
    if rank == 0:
        for k, v in sd0.items():
            res["sd0/" + k] = v.cpu().numpy()
    np.savez(out, **res)


def case_oneshot(rank, world, out):
    """exchange='oneshot' (csrc/oneshot.hip): the library's one-shot all-to-all over IPC-mapped inboxes.  Raw exchanges of known
    payloads (odd lengths, lengths that are not a multiple of the 4096-double chunk, 25 exchanges back to back through both parities
    of the double-buffered inbox), then the sharded k-means on the goldens."""
    import ctypes
    from video_similarity_search_amd import _lib
    from video_similarity_search_amd._lib import call, ptr, stream
    from video_similarity_search_amd.clustering import kmeans_hip as kh
    pg = dist.group.WORLD
    dev = torch.device("cuda", torch.cuda.current_device())
    max_n = 70001
    comm = kh._slic_oneshot(pg, dev, max_n)
    res = {}
    ok = True
    rng = np.random.default_rng(5)
    sizes = [1, 2, 3, 4095, 4096, 4097, 8192, 70001, 12345, 70000] + [int(v) for v in rng.integers(1, max_n, 15)]
    for it, n in enumerate(sizes):
        # every rank can rebuild every rank's payload: integers below 2^20 times a rank-specific scale, exact in fp64
        parts = [np.random.default_rng(1000 * it + r).integers(-2 ** 20, 2 ** 20, n).astype(np.float64) * (r + 1) for r in range(world)]
        buf = torch.from_numpy(parts[rank]).cuda()
        call("slic_allreduce_oneshot_f64", comm, ptr(buf), n, stream())
        want = parts[0].copy()
        for r in range(1, world):
            want += parts[r]
        torch.cuda.synchronize()
        call("slic_oneshot_check", comm)
        ok = ok and np.array_equal(buf.cpu().numpy(), want)
    res["raw_ok"] = ok
    res["n_exchanges"] = len(sizes)
    # 300 exchanges back to back with NO host synchronisation between them (the k-means loop runs one iteration ahead of its reads): the
    # double buffering of the inboxes by exchange parity must hold when a fast rank is a whole exchange ahead of a slow one.  Every rank
    # refills the buffer in stream order, a checksum accumulates in stream order; rank 1 dawdles (extra work on its stream) every few steps.
    n = 40001
    buf = torch.empty(n, dtype=torch.float64, device="cuda")
    acc = torch.zeros(n, dtype=torch.float64, device="cuda")
    junk = torch.randn(2048, 2048, device="cuda")
    for it in range(300):
        buf.fill_(float(rank + 1 + (it % 7)))
        if rank == 1 and it % 5 == 0:
            junk = torch.tanh(junk) + 1.0                           # ~100 us of unrelated work in front of this rank's push
        call("slic_allreduce_oneshot_f64", comm, ptr(buf), n, stream())
        acc += buf
    torch.cuda.synchronize()
    call("slic_oneshot_check", comm)
    want = sum(sum(r + 1 + (it % 7) for r in range(world)) for it in range(300))
    res["stress_ok"] = bool((acc == float(want)).all().item())
    info = (ctypes.c_int * 4)()
    call("slic_oneshot_info", comm, info)                          # world, rank, memory kind, exchanges issued
    res["info"] = np.array(list(info))
    # the sharded Lloyd iteration with this exchange: labels of every iteration bit-equal to the oracle with n_shards = -W (the parent compares)
    from video_similarity_search_amd.clustering.kmeans_hip import KMeans
    for name in ("clustered_empty", "d128", "unstructured"):
        g = dict(np.load(os.path.join(HERE, "golden", f"kmeans_{name}.npz")))
        X, init = g["X"], g["init"]
        N = len(X)
        per = (N + world - 1) // world
        shard = torch.from_numpy(X[rank * per:(rank + 1) * per]).cuda()
        km = KMeans(n_clusters=init.shape[0], init=init, n_init=1, process_group=pg, trace=True, exchange="oneshot").fit(shard)
        res[f"{name}/labels_local"] = km.labels_
        res[f"{name}/centers"] = km.cluster_centers_
        res[f"{name}/n_iter"] = km.n_iter_
        res[f"{name}/inertia"] = km.inertia_
        res[f"{name}/trace_local"] = km.trace_
        res[f"{name}/comm"] = np.array(km.communicator_kind_)
    np.savez(out, **res)


def case_oneshot_timeout(rank, world, out):
    """a lost peer: rank 1 sets the exchange up with rank 0 and then never pushes.  Rank 0's exchange kernel gives up after
    SLIC_COMM_TIMEOUT_MS (set to 2 s by the test), completes, and slic_oneshot_check reports SLIC_ETIMEOUT — no hang; later
    exchanges on the communicator are refused."""
    import time
    from video_similarity_search_amd import _lib
    from video_similarity_search_amd._lib import call, ptr, stream
    from video_similarity_search_amd.clustering import kmeans_hip as kh
    pg = dist.group.WORLD
    dev = torch.device("cuda", torch.cuda.current_device())
    comm = kh._slic_oneshot(pg, dev, 10000)
    res = dict(rank=rank)
    if rank == 0:
        buf = torch.ones(10000, dtype=torch.float64, device="cuda")
        t0 = time.time()
        call("slic_allreduce_oneshot_f64", comm, ptr(buf), 10000, stream())
        torch.cuda.synchronize()
        res["seconds"] = time.time() - t0
        try:
            call("slic_oneshot_check", comm)
            res["raised"] = False
        except _lib.SlicError as e:
            res["raised"] = True
            res["msg"] = np.array(str(e))
        try:
            call("slic_allreduce_oneshot_f64", comm, ptr(buf), 10000, stream())
            res["refused_after"] = False
        except _lib.SlicError:
            res["refused_after"] = True
    np.savez(out, **res)


def case_launch(rank, world, out):
    raise SystemExit("case_launch is driven by the test itself (misc.distributed_helper.launch_processes)")


def main():
    case, out_dir = sys.argv[1], sys.argv[2]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SLIC_TEST_SAME_GPU") == "1":
        # every rank on GPU 0 (how a one-GPU box runs two ranks of the one-shot exchange): RCCL cannot put two ranks on one device, so the
        # process group — which only carries set-up data and small host-staged collectives here — is gloo
        torch.cuda.set_device(0)
        dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl")
    rank, world = dist.get_rank(), dist.get_world_size()
    try:
        globals()["case_" + case](rank, world, os.path.join(out_dir, f"{case}_r{rank}.npz"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
