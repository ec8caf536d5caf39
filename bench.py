#!/usr/bin/env python3
"""
bench.py — headline benchmark of the MI355X hot path (contract: the round driver).

    python bench.py --gpus N --steps K --warmup W
        N > 1: one rank per GPU over RCCL — either already under `python -m torch.distributed.run --nproc-per-node N ...`
        (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or typed plainly, in which case this process
        starts that launcher as a child before touching the GPU and relays rank 0's JSON line.

Step = one pass of the training hot path over one synthetic batch: R3D-18 (models/resnet.py depth 18, the SLIC
encoder) forward + backward + NT-Xent ('noise_contrastive') + SGD(lr .1, momentum .5) on a
32 x 3 x 16 x 112 x 112 fp32 batch (16 anchors || 16 positives) per GPU — BASELINE.json configs[1].
Inputs are resident in HBM before the timed region.  N > 1: DistributedDataParallel over RCCL (gradient
all-reduce inside the step), weak scaling (per-GPU batch fixed), value = clips of all ranks / max-over-ranks time.

One JSON line on rank 0 with: the contract keys, `roofline` for the dominant kernel (the 64->64 3x3x3
gather-GEMM, HIP events around its launches inside the timed steps), `cpu_baseline` (the CPU oracle timed on
this host, bounded sample), and `secondary` (k-means embeddings/s at 100k x 512, K = 500 — the second half of
BASELINE.json's metric — with its own roofline and CPU baseline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

FP32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 MFMA peak
GFLOP_PER_CLIP_TRAIN = 248.9         # fwd 85.17 + bwd 163.7 (2*MAC, conv + linear)
R3D18_KW = dict(hidden_layer=2048, out_dim=128, num_classes=101, n_input_channels=3, shortcut_type='B',
                conv1_t_size=7, conv1_t_stride=1, no_max_pool=True, widen_factor=1.0, projection_head=True,
                predict_temporal_ds=False, spatio_temporal_attention=False, classifier=False, dropout=None)


class _Skip(Exception):
    pass


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_model(seed=7):
    """R3D-18 with the reference's init rules (models/resnet.py:203-210), random-init (no checkpoints offline)"""
    import contextlib
    import io
    from video_similarity_search_amd.models import generate_model
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        m = generate_model(18, **R3D18_KW)
    sd = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
    return m, sd


def cpu_baseline_encoder(sd, seconds_budget=25.0):
    """the CPU oracle (oracle/encoder.py: the reference's math in PyTorch-CPU ops) on this host's cores:
    fwd + bwd + NT-Xent + SGD at B = 4 (bounded sample of the same workload)"""
    from oracle import encoder as oe
    B = 4
    rng = np.random.default_rng(7)
    x = torch.from_numpy(rng.standard_normal((B, 3, 16, 112, 112)).astype(np.float32))
    t = oe.to_torch(sd, requires_grad=True)
    params = {k: v for k, v in t.items() if v.requires_grad}
    bufs = {}

    def step():
        emb = oe.encoder_forward(t, x, training=True)
        loss = oe.ntxent_loss(emb)
        grads = dict(zip(params, torch.autograd.grad(loss, list(params.values()))))
        oe.sgd_step(params, grads, bufs)
    step()                                   # warm-up
    n, t0 = 0, time.time()
    while n < 2 or (time.time() - t0 < seconds_budget * 0.5 and n < 6):
        step()
        n += 1
    dt = (time.time() - t0) / n
    return dict(value=B / dt, unit="clips/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle/encoder.py R3D-18 fwd+bwd+NT-Xent+SGD, B={B} x 3x16x112x112 fp32, {n} steps after 1 warm-up, "
                       f"{dt:.2f} s/step, torch {torch.__version__} CPU, os.cpu_count()={os.cpu_count()}")


def _exec_factor(p):
    """share of the direct form's multiplies a plan's forward / data-gradient kernel executes: Winograd F(4,3) x F(2,3) over (W, H)
    a third, F(4,3) along W a half"""
    return (1.0 / 3.0) if getattr(p, "wino2", False) else (0.5 if p.wino else 1.0)


def step_flops(eng, B):
    """(algorithmic, executed) matrix FLOPs of one training step on B clips, from the engine's own plans: forward + data gradient
    (every layer but the stem) + weight gradient of every convolution / linear layer.  A Winograd F(4,3) launch executes half of
    the direct form's multiplies."""
    plans = [(eng.stem, False)]
    for blk, p1, p2, pd in eng.blocks:
        plans += [(p1, True), (p2, True)] + ([(pd, True)] if pd is not None else []) + ([(eng.p3[blk], True)] if blk in eng.p3 else [])
    if eng.net.projection_head:
        plans += [(eng.fc1, True), (eng.fc2, True)]
    alg = exe = 0.0
    for p, has_dgrad in plans:
        f = 2.0 * B * float(np.prod(p.out_dims)) * p.N * p.C * p.ntaps
        alg += f * (2 + has_dgrad)
        exe += f * (_exec_factor(p) * (1 + has_dgrad) + ((1.0 / 3.0) if getattr(p, "wino2_wgrad", False) else 0.5 if p.wino_wgrad else 1.0))
    return alg, exe


def fwd_flops(eng, B):
    """(algorithmic, executed) matrix FLOPs of one forward pass on B clips"""
    plans = [eng.stem] + [p for blk, p1, p2, pd in eng.blocks for p in (p1, p2, pd, eng.p3.get(blk)) if p is not None]
    if eng.net.projection_head:
        plans += [eng.fc1, eng.fc2]
    alg = exe = 0.0
    for p in plans:
        f = 2.0 * B * float(np.prod(p.out_dims)) * p.N * p.C * p.ntaps
        alg += f
        exe += f * _exec_factor(p)
    return alg, exe


def kmeans_secondary(rank, world, pg, run_cpu):
    """k-means Lloyd throughput at BASELINE configs[2]: 100k x 512, K = 500, explicit init, tol = 0, fixed 20 iterations:
    embeddings/s = N * iters / wall (assign + update + status sync + collective).  With a process group the rows are sharded
    over the ranks and an iteration is two foreign calls around ONE RCCL collective (fp64 all-reduce of the payload
    [K*D sums | K counts | n_changed]); without one (plain N = 1) the whole matrix is on the one GPU and an iteration is one
    fused foreign call."""
    from video_similarity_search_amd.clustering import KMeans
    N, D, K, iters = 100000, 512, 500, 20
    rng = np.random.default_rng(1)
    X = rng.standard_normal((N, D)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    init = X[rng.choice(N, K, replace=False)].copy()
    per = (N + world - 1) // world
    Xd = torch.from_numpy(X[rank * per:(rank + 1) * per]).cuda()
    km = KMeans(n_clusters=K, init=init, n_init=1, max_iter=iters, tol=0.0, fixed_iters=True, process_group=pg)
    # warm-up: FOUR whole fits (the first also allocates workspaces).  A fit is 10 ms of GPU work and the chip's clock is still ramping
    # through the first tens of milliseconds after the host-side set-up above: the E-step of a first fit runs 445 -> 415 us, of the
    # fifth 392 (rocprofv3 trace, round 4) — one warm-up fit measured the ramp, not the kernels
    per_fit = []
    for _ in range(4):
        km.fit(Xd)
        per_fit.append(km.lloyd_seconds_ / iters * 1e3)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    # metric 2 is the wall time of the Lloyd phase — assign + update + convergence test (+ collective) —
    # which KMeans brackets itself with device syncs (lloyd_seconds_); the whole fit (centring, tolerance, the final
    # relabelling E-step, inertia) is reported beside it.  TWO timed fits, the mean of their Lloyd phases.
    dts, dt_fits = [], []
    for _ in range(2):
        t0 = time.time()
        km.fit(Xd)
        torch.cuda.synchronize()
        dt_fits.append(time.time() - t0)
        dts.append(km.lloyd_seconds_)
        assert km.lloyd_iters_ == iters
        per_fit.append(km.lloyd_seconds_ / iters * 1e3)
    dt, dt_fit = sum(dts) / len(dts), sum(dt_fits) / len(dt_fits)
    if world > 1:
        tt = torch.tensor([dt, dt_fit], device="cuda")
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt, dt_fit = (float(v) for v in tt.tolist())
    layout = ("rows sharded over the ranks of an RCCL group, ONE fp64 all-reduce of [K*D sums | K counts | n_changed] per iteration "
              "between slic_kmeans_lloyd_local and slic_kmeans_lloyd_global"
              if pg is not None else "whole matrix on one GPU, no process group (one fused foreign call per iteration)")
    flops_iter = 2.0 * N * K * D
    out = dict(metric="k-means embeddings/sec 100kx512 K=500", value=N * iters / dt, unit="embeddings/s",
               ms_per_iter=dt / iters * 1e3, iters=iters, n_gpus=world, whole_fit_seconds=dt_fit,
               ms_per_iter_of_each_fit=[round(v, 4) for v in per_fit], protocol="4 warm-up fits, mean of 2 timed fits of 20 iterations",
               whole_iteration_frac_of_fp32_mfma=flops_iter / (dt / iters) / 1e12 / (FP32_MFMA_PEAK_TFLOPS * world),
               config=dict(workload=f"Lloyd, N=100000 D=512 K=500 fp32, explicit init, tol=0, 20 fixed iterations; {layout}"))
    if pg is not None:
        out["exchange"] = dict(kind=km.exchange, communicator=km.communicator_kind_, payload_bytes_per_rank=km.payload_bytes_,
                               collectives_per_iteration=1)
        # the same iteration WEAK-scaled (100k rows on every GPU, N = 100k x ranks): per-GPU work fixed, so the curve shows what the one
        # collective per iteration costs; the strong-scaled row above divides 100k rows over the ranks (per-rank E-step of ~50 us at 8)
        Xw = torch.from_numpy(X).cuda() if world > 1 else Xd
        kmw = KMeans(n_clusters=K, init=init, n_init=1, max_iter=iters, tol=0.0, fixed_iters=True, process_group=pg)
        for _ in range(3):
            kmw.fit(Xw)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        kmw.fit(Xw)
        torch.cuda.synchronize()
        dtw = kmw.lloyd_seconds_
        if world > 1:
            tw = torch.tensor([dtw], device="cuda")
            torch.distributed.all_reduce(tw, op=torch.distributed.ReduceOp.MAX)
            dtw = float(tw.item())
        out["weak_scaled"] = dict(value=N * world * iters / dtw, unit="embeddings/s", ms_per_iter=dtw / iters * 1e3, n_gpus=world,
                                  rows_per_gpu=N, scaling="weak")
        del Xw, kmw
    # E-step kernel alone (dominant kernel of this path): HIP events on the launch stream
    from video_similarity_search_amd.clustering.kmeans_hip import HipKernels
    k = HipKernels()
    n_loc = Xd.shape[0]
    C = torch.from_numpy(init).cuda()
    cn = torch.empty(K, device="cuda")
    lab = torch.empty(n_loc, dtype=torch.int32, device="cuda")
    k.cnorm(C, cn)
    Xp, Cp = torch.empty_like(Xd), torch.empty_like(C)
    k.permute_k8(Xd, Xp)
    k.permute_k8(C, Cp)
    k.assign_perm(Xp, Cp, cn, lab, None, None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        k.assign_perm(Xp, Cp, cn, lab, None, None)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * n_loc * K * D
    out["roofline"] = dict(bound="mfma", achieved=flops / (ms * 1e-3) / 1e12, peak=FP32_MFMA_PEAK_TFLOPS,
                           unit="TFLOP/s", frac=flops / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, traffic=None,
                           kernel="km_assign_creg<16,4> (centroids in registers, points through a 4-stage LDS ring) + km_combine", ms_per_launch=ms,
                           algorithmic_flops_per_launch=flops)
    if run_cpu:
        from oracle import kmeans as ok
        mean = ok.col_mean(X)
        t0 = time.time()
        ok.lloyd(X - mean, init - mean, max_iter=3, tol_abs=0.0, fixed_iters=True)
        dtc = time.time() - t0
        out["cpu_baseline"] = dict(value=N * 4 / dtc, unit="embeddings/s", cores=ok.num_threads(), kind="port",
                                   sample="oracle/kmeans_oracle.c (OpenMP, bit-exact fmaf chains) 3 Lloyd iterations + final "
                                          f"E-step on the same 100k x 512, K=500 data: {dtc:.2f} s")
        try:
            # sklearn.cluster.KMeans itself (third-party code the reference calls, cluster_masks.py:70-71),
            # same explicit init, tol = 0, a few fixed iterations, on this host's cores
            import warnings
            from sklearn.cluster import KMeans as SkKMeans
            from threadpoolctl import threadpool_info
            it_sk = 5
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                SkKMeans(n_clusters=K, init=init, n_init=1, max_iter=1, tol=0.0, algorithm="lloyd").fit(X[:20000])   # warm-up
                t0 = time.time()
                sk = SkKMeans(n_clusters=K, init=init, n_init=1, max_iter=it_sk, tol=0.0, algorithm="lloyd").fit(X)
                dts = time.time() - t0
            nthr = max([int(i.get("num_threads", 1)) for i in threadpool_info()] + [1])
            out["cpu_baseline_sklearn"] = dict(
                value=N * sk.n_iter_ / dts, unit="embeddings/s", cores=nthr, kind="third-party",
                sample=f"sklearn {__import__('sklearn').__version__} KMeans(init=ndarray, n_init=1, max_iter={it_sk}, tol=0, lloyd).fit on the "
                       f"same data: {dts:.2f} s wall for {sk.n_iter_} iterations (incl. its centring and final E-step), os.cpu_count()={os.cpu_count()}")
        except Exception as e:
            out["cpu_baseline_sklearn"] = dict(error=repr(e))
    return out


def kmeans_oneshot_row(rank, world, pg):
    """the strong-scaled k-means row of kmeans_secondary with exchange='oneshot'"""
    from video_similarity_search_amd.clustering import KMeans
    N, D, K, iters = 100000, 512, 500, 20
    rng = np.random.default_rng(1)
    X = rng.standard_normal((N, D)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    init = X[rng.choice(N, K, replace=False)].copy()
    per = (N + world - 1) // world
    Xd = torch.from_numpy(X[rank * per:(rank + 1) * per]).cuda()
    km = KMeans(n_clusters=K, init=init, n_init=1, max_iter=iters, tol=0.0, fixed_iters=True, process_group=pg, exchange="oneshot")
    for _ in range(4):
        km.fit(Xd)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dts = []
    for _ in range(2):
        km.fit(Xd)
        torch.cuda.synchronize()
        dts.append(km.lloyd_seconds_)
    dt = sum(dts) / len(dts)
    if world > 1:
        tt = torch.tensor([dt], device="cuda")
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    return dict(metric="k-means embeddings/sec 100kx512 K=500, rows sharded, one-shot exchange", value=N * iters / dt, unit="embeddings/s",
                ms_per_iter=dt / iters * 1e3, n_gpus=world, exchange=dict(kind=km.exchange, communicator=km.communicator_kind_,
                                                                          payload_bytes_per_rank=km.payload_bytes_))


def _time_events(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def measure_traffic():
    """HBM bytes per launch of the dominant kernel (conv_wino2p_kernel<false, 64>: layer1's forward launches) from rocprofv3 PMC passes over a
    CHILD `bench.py --steps 2 --warmup 1`: FETCH_SIZE and WRITE_SIZE in separate passes; gfx950's FETCH_SIZE reports half the bytes of a wide
    (16 B / lane) coalesced read (MI355X_MICROARCH.md, HBM section), so bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB — an upper bound for the
    32-byte pixel pieces.  Algorithmic: 0.82 GB per forward launch (x read once, z written once)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        raise RuntimeError("rocprofv3 not on PATH")
    kern = "conv_wino2p_kernel<false, 64>" if os.environ.get("SLIC_WINO2_PERSIST", "1") != "0" else "conv_wino2_kernel"
    grid = (256 if "wino2p" in kern else 3136) * 512
    out = {}
    tmp = tempfile.mkdtemp(prefix="slic_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--no-traffic"]
            env = dict(os.environ, TMPDIR="/tmp", SLIC_BENCH_CHILD="1")
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120, check=True)
            f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
            per = {}
            for r in csv.DictReader(open(f[-1])):
                if kern in r["Kernel_Name"] and r["Counter_Name"] == counter and int(r.get("Grid_Size", r.get("Grid_Size_X", 0))) == grid:
                    per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
            if not per:
                raise RuntimeError(f"no {kern} dispatch in the {counter} pass")
            out[counter + "_KiB_mean"] = sum(per.values()) / len(per)
            out[counter + "_launches"] = len(per)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out["kernel"] = kern
    out["hbm_bytes_per_launch"] = (2.0 * out["FETCH_SIZE_KiB_mean"] + out["WRITE_SIZE_KiB_mean"]) * 1024.0
    out["algorithmic_bytes_per_launch"] = 2.0 * 32 * 16 * 56 * 56 * 64 * 4
    out["note"] = ("rocprofv3 --kernel-trace --pmc <counter> over a child `bench.py --steps 2 --warmup 1`, one pass per counter; "
                   "(2 x FETCH_SIZE + WRITE_SIZE) KiB per MI355X_MICROARCH.md's gfx950 correction (upper bound for the 32-byte pixel pieces)")
    return out


def retrieval_secondary(run_cpu):
    """BASELINE configs[4] (iic_retrieve_clips.py:275-314): 10k x 512 queries vs 100k x 512 gallery, cosine top-50, one GPU;
    inputs resident in HBM; normalise + fused similarity/top-k + merge per call"""
    from video_similarity_search_amd.evaluate import cosine_topk
    Nq, Ng, D, k = 10000, 100000, 512, 50
    rng = np.random.default_rng(5)
    Qh = rng.standard_normal((Nq, D)).astype(np.float32)
    Gh = rng.standard_normal((Ng, D)).astype(np.float32)
    Q, G = torch.from_numpy(Qh).cuda(), torch.from_numpy(Gh).cuda()
    t = _time_events(lambda: cosine_topk(Q, G, k=k), 5)
    fl = 2.0 * Nq * Ng * D
    # the label follows the library's own choice for this shape (slic_cosine_topk_plan), not a constant
    import ctypes
    from video_similarity_search_amd import _lib
    plan = (ctypes.c_int * 6)()
    _lib.check(_lib.load().slic_cosine_topk_plan(Nq, Ng, D, k, plan), "slic_cosine_topk_plan")
    nk = 16 if D > 256 else 8 if D > 128 else 4
    if plan[0]:
        kern = (f"topk_collect_qreg<{nk},4> (threshold -> collect -> select: + topk_partial_qreg sample pass of {plan[1]} x {plan[2]} rows, "
                f"topk_thresholds, topk_select; whole call incl. row normalisation)")
    else:
        kern = f"topk_partial_qreg<{nk},4> + topk_merge_kernel (streaming heaps; whole call incl. row normalisation)"
    out = dict(metric="retrieval queries/sec, 10k x 512 vs 100k x 512 cosine top-50", value=Nq / t, unit="queries/s",
               ms=t * 1e3, roofline=dict(bound="mfma", achieved=fl / t / 1e12, peak=FP32_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                                         frac=fl / t / 1e12 / FP32_MFMA_PEAK_TFLOPS, traffic=None,
                                         kernel=kern, algorithm="collect" if plan[0] else "streaming",
                                         algorithmic_flops_per_launch=fl))
    if run_cpu:
        # the reference's own arithmetic on the host (sklearn cosine_distances = normalise + GEMM; then top-k): NumPy
        # Q^ G^T + argpartition + sort of the kept k, on a bounded sample of the queries
        ns = 500
        t0 = time.time()
        qn = Qh[:ns] / np.linalg.norm(Qh[:ns], axis=1, keepdims=True)
        gn = Gh / np.linalg.norm(Gh, axis=1, keepdims=True)
        Dm = 1.0 - qn @ gn.T
        idx = np.argpartition(Dm, k, axis=1)[:, :k]
        part = np.take_along_axis(Dm, idx, axis=1)
        idx = np.take_along_axis(idx, np.argsort(part, axis=1), axis=1)
        dtc = time.time() - t0
        gi = cosine_topk(Q[:ns], G, k=k)[0].cpu().numpy()
        same = float(np.mean([len(set(a) & set(b)) / k for a, b in zip(gi, idx)]))
        out["cpu_baseline"] = dict(value=ns / dtc, unit="queries/s", cores=os.cpu_count(), kind="port",
                                   sample=f"NumPy normalise + sgemm + argpartition/sort top-{k}: {ns} of the 10k queries vs the full "
                                          f"100k gallery in {dtc:.2f} s (gallery normalisation included once); top-{k} set overlap with the GPU result {same:.4f}")
    return out


def nce_secondary():
    """SURVEY.md §8 A4 at B = 32, K = 1024, D = 128, n_data = 100k: NCEAverage + 2 x NCESoftmaxLoss forward + backward + bank update"""
    from video_similarity_search_amd.loss.NCE_loss import NCEAverage, NCESoftmaxLoss
    B, K, D, n_data = 32, 1024, 128, 100000
    nce = NCEAverage(D, n_data, K).cuda()
    crit = NCESoftmaxLoss().cuda()
    l = torch.randn(B, D, device="cuda", requires_grad=True)
    ab = torch.randn(B, D, device="cuda", requires_grad=True)
    y = torch.randint(0, n_data, (B,), device="cuda")

    def nce_step_modules():
        o1, o2 = nce(l, ab, y)
        (crit(o1) + crit(o2)).backward()

    def nce_step():                     # what contrastive_train_epoch runs: the same step as three launches
        nce.softmax_loss(l, ab, y)[0].backward()

    t_modules = _time_events(nce_step_modules, 20)
    t = _time_events(nce_step, 20)
    bytes_alg = 2 * B * (K + 1) * D * 4 + 2 * B * D * 4 * 2
    return dict(metric="memory-bank NCE step (fwd + bwd + bank update), B=32 K=1024 D=128", ms=t * 1e3, ms_module_by_module=t_modules * 1e3,
                #  this row is HOST-issue-bound: 37 us of kernels inside ~0.19 ms of Python autograd + index draw per step.  The
                # fraction of the wall time would describe the host, not a kernel — the roofline object is therefore priced on the KERNEL time of the
                # three launches (profiles/: 16 + 11 + 10 us), and the wall time is printed beside it as what it is
                bound_in_practice="host issue (Python autograd node + index draw), not the GPU",
                roofline=dict(bound="hbm", achieved=bytes_alg / 37e-6 / 1e9, peak=8000.0, unit="GB/s", frac=bytes_alg / 37e-6 / 1e9 / 8000.0,
                              traffic=None, algorithmic_bytes_per_step=bytes_alg, priced_on="kernel time of the three launches (37 us, rocprofv3 trace under profiles/), not the wall time per step",
                              frac_of_wall_time=bytes_alg / t / 1e9 / 8000.0,
                              note="NCEAverage.softmax_loss: three launches (scores + cross-entropy pieces, bank update + loss, backward) over "
                                   "33.6 MB of gathered bank rows — 16 + 11 + 10 us of kernel time (profiles/README.md); the wall time per step "
                                   "timed here is the host's issue cost (Python autograd node + index draw); the module-by-module form "
                                   "(a dozen launches) is timed beside it"))


def self_launch(n):
    """`python bench.py --gpus N` typed plainly: start the N ranks as a CHILD `torch.distributed.run` (one process per GPU,
    rendezvous on 127.0.0.1) and relay rank 0's JSON line.  Nothing in this parent has touched the GPU yet (importing torch
    does not initialise HIP), and the parent never exec()s — it waits and exits with the child's code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    argv = [a for a in sys.argv[1:] if a != "--self-launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    log("bench.py: launching", " ".join(cmd))
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in p.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        else:
            log(out)
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        log("bench.py: the ranks exited cleanly but rank 0 printed no JSON line")
        rc = 3
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)          # >= 20 timed steps ...
    ap.add_argument("--warmup", type=int, default=5)         # ... after >= 5 warm-up steps
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU per step (16 anchors || 16 positives)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the child rocprofv3 --pmc passes that fill roofline.traffic")
    ap.add_argument("--quick", action="store_true", help="secondary = the k-means row only (tests)")
    ap.add_argument("--force-dist", action="store_true", help="init the process group / DDP / sharded k-means even at world size 1 (path test)")
    ap.add_argument("--self-launch", action="store_true", help="go through the child torch.distributed.run launcher even for --gpus 1 (path test)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.self_launch):
        sys.exit(self_launch(args.gpus))
    if args.gpus != world:
        log(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    pg = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # (with NCCL_DEBUG=VERSION in the environment — the GPU boxes set it — RCCL prints its version banner to stdout at the first
        #  communicator; NCCL_DEBUG_FILE does not move it.  The JSON line is the only stdout line that starts with '{'.)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import datetime
        # RCCL on ROCm.  A bounded timeout: should one rank fail inside a secondary row while its peers wait in a collective,
        # the job aborts after five minutes instead of hanging (the headline line is printed before the secondary rows start)
        torch.distributed.init_process_group(backend="nccl", timeout=datetime.timedelta(minutes=5))
        pg = torch.distributed.group.WORLD
        # The sharded k-means row's ONE all-reduce per iteration goes through torch's RCCL process group (the product default:
        # its collectives carry the timeout above); SLIC_KMEANS_COMM=slic selects the library's own communicator
        # (slic_allreduce_f64, include/slic_hip.h), whose waits are bounded by SLIC_COMM_TIMEOUT_MS.

    from video_similarity_search_amd import _lib
    _lib.check(_lib.load().slic_device_check(), "slic_device_check")
    from video_similarity_search_amd.loss import OnlineTripletLoss

    model, sd = build_model()
    model = model.cuda().train()
    net = model
    if use_dist:
        ddp_kw = dict(gradient_as_bucket_view=os.environ.get("SLIC_DDP_BUCKET_VIEW", "1") != "0",
                      bucket_cap_mb=int(os.environ.get("SLIC_DDP_BUCKET_MB", "25")))
        # SLIC_DDP_FAST=1 (default): misc.distributed_helper.data_parallel — the same DistributedDataParallel wrapper with its per-step copies
        # removed (flat buffer broadcast, gradients written into the bucket views, ReduceOp.AVG hook); 0: the reference's plain call
        # (online_train.py:485-494)
        if os.environ.get("SLIC_DDP_FAST", "1") != "0":
            from video_similarity_search_amd.misc.distributed_helper import data_parallel
            try:
                model = data_parallel(model, local_rank, bucket_cap_mb=ddp_kw["bucket_cap_mb"])
            except Exception as e:                                # never lose a multi-GPU run to the fast wrapper: the reference's call still works
                log(f"bench.py: rank {rank}: data_parallel failed ({e!r}); falling back to the plain DistributedDataParallel call")
                model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[local_rank], **ddp_kw)
        else:
            model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], **ddp_kw)
    crit = OnlineTripletLoss(0.2, 'cosine')
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.5)
    B = args.batch
    rng = np.random.default_rng(7 + rank)
    x = torch.from_numpy(rng.standard_normal((B, 3, 16, 112, 112)).astype(np.float32)).cuda()   # resident in HBM
    labels = torch.arange(B // 2).repeat(2).cuda()

    def step():
        emb = model(x)
        loss, _ = crit(emb, labels, sampling_strategy='noise_contrastive')
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    # dominant kernel (largest share of a step's kernel time): conv_wino2p_kernel (round 6: the persistent form of conv_wino2_kernel) — Winograd F(4,3) along W x F(2,3) along H — on the 64->64
    # 3x3x3 layers (layer1: 4 forward + 4 data-gradient launches per step, identical M x N x K).  Every such launch of the timed steps is
    # bracketed with HIP events on the launch stream.  `roofline.achieved` / `frac` count the fp32 MFMA FLOPs the launch EXECUTES (a third of
    # the direct form's 2 M N K) over its duration: a fraction of the pipe's peak, <= 1; the direct form's count is printed beside it.
    eng = net._engine(x)
    l1 = [p for (_, p1, p2, _) in eng.blocks[:2] for p in (p1, p2)]
    for p in l1:
        p.prof = []
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.time() - t0
    dist_info = dict(initialised=bool(use_dist), world_size_env=world)
    if use_dist:
        # self-verification of a multi-GPU run: how many ranks RCCL really reduced over (a sum of ones), the library
        # version, every rank's own step time, and what DistributedDataParallel did with the gradients
        ones = torch.ones(1, device="cuda")
        torch.distributed.all_reduce(ones)
        per_rank = [torch.zeros(1, dtype=torch.float64, device="cuda") for _ in range(world)]
        torch.distributed.all_gather(per_rank, torch.tensor([dt / args.steps * 1e3], dtype=torch.float64, device="cuda"))
        per_rank = [float(v.item()) for v in per_rank]
        ld = model._get_ddp_logging_data()
        bsz = str(ld.get("rebuilt_bucket_sizes") or ld.get("bucket_sizes") or "")
        dist_info.update(backend=torch.distributed.get_backend(), rccl_ranks_seen=int(round(float(ones.item()))),
                         rccl_version=".".join(str(v) for v in torch.cuda.nccl.version()),
                         ms_per_step_per_rank=[round(v, 4) for v in per_rank], ms_per_step_min=min(per_rank), ms_per_step_max=max(per_rank),
                         ddp=dict(buckets=len([b for b in bsz.split(",") if b.strip()]), bucket_sizes_bytes=bsz,
                                  bucket_cap_mb=ddp_kw["bucket_cap_mb"], gradient_as_bucket_view=bool(ld.get("gradient_as_bucket_view", ddp_kw["gradient_as_bucket_view"])),
                                  backend_name=ld.get("backend_name"), gradient_bytes=int(sum(p.numel() for p in net.parameters()) * 4),
                                  wrapper=("misc.distributed_helper.data_parallel" if hasattr(model, "slic_ddp") else "torch.nn.parallel.DistributedDataParallel (plain)"),
                                  slic_ddp=getattr(model, "slic_ddp", None)),
                         devices=[torch.cuda.get_device_name(local_rank)], hsa_ipc_legacy=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
    if world > 1:
        tt = torch.tensor([dt], device="cuda")
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    lossv = float(loss.item())
    # layer1's FORWARD launches: the same kernel, grid and K as its data-gradient launches, and nothing shares the GPU with them (the
    # backward's weight gradients run on a side stream beside the data gradients, so a data-gradient launch's duration describes two
    # kernels' share of the chip; it is reported beside the forward figure, not used for the roofline)
    ev = [a.elapsed_time(b) for p in l1 for (a, b, tag) in p.prof if tag == "fwd"]
    ev_dg = [a.elapsed_time(b) for p in l1 for (a, b, tag) in p.prof if tag == "dgrad"]
    for p in l1:
        p.prof = None
    ms_k = float(np.mean(ev))
    M = B * 16 * 56 * 56
    flops_launch = 2.0 * M * 64 * 1728                                 # ALGORITHMIC: the direct convolution's 2 M N K
    wino = bool(getattr(l1[0], "wino", False))
    wino2 = bool(getattr(l1[0], "wino2", False))
    executed = flops_launch * _exec_factor(l1[0])                    # F(4,3): 6 multiplies where the direct form has 12; x F(2,3): 4
    # `achieved` / `frac` = what the matrix pipe really EXECUTES (fp32 MFMA FLOPs of the launch / its duration / the pipe's peak):
    # a fraction of a hardware peak, <= 1 by construction.  The direct-form FLOP count of the same convolution over the same time is
    # reported beside it as algorithmic_tflops; the two differ by the Winograd factor (algorithmic_speedup_vs_direct).
    ach = executed / (ms_k * 1e-3) / 1e12
    alg_step, exe_step = step_flops(eng, B)

    # HBM traffic of the dominant kernel comes from PMC counters, which need rocprofv3 around the process: the value below is
    # the one committed with the profile of the same command (profiles/), NOT measured in this run — `traffic` stays null
    traffic_prof, traffic_src, trace_ms_prof = None, None, None
    for name in (("r06_pmc_conv_wino.json", "r05_pmc_conv_wino.json", "r04_pmc_conv_wino.json", "r03_pmc_conv_wino.json") if wino else
                 ("r02_pmc_conv_gemm_dma.json", "r01_pmc_conv_gemm_dma.json")):
        pmc = os.path.join(ROOT, "profiles", name)
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                traffic_prof, traffic_src, trace_ms_prof = pj.get("hbm_bytes_per_launch"), "profiles/" + name, pj.get("rocprof_trace_avg_ms")
                break
            except Exception:
                pass

    try:
        metric_name = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]     # the exact string of BASELINE.json
    except Exception:
        metric_name = "clips/sec R3D-18+NCE (1/2/4/8 GPU); k-means embeddings/sec 100k\u00d7512"
    res = dict(metric=metric_name,
               value=world * B * args.steps / dt, unit="clips/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
               dtype="f32", data="synthetic", distributed=dist_info,
               config=dict(workload="R3D-18 (3D-ResNet-18, 34.52 M params, no maxpool, 7^3 stem) fwd+bwd + NT-Xent "
                                    "(noise_contrastive, T=0.5) + SGD(lr .1, mom .5); per-GPU batch "
                                    f"{B} x 3x16x112x112 fp32 = {B//2} anchors || {B//2} positives; BASELINE configs[1]",
                           global_batch=world * B, parallelism=f"dp{world}", final_loss=lossv),
               roofline=dict(bound="mfma", achieved=ach, peak=FP32_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                             frac=ach / FP32_MFMA_PEAK_TFLOPS, traffic=None, traffic_from_profile=traffic_prof,
                             traffic_source=traffic_src,
                             kernel=((("conv_wino2p_kernel<false> — the PERSISTENT form (256 workgroups walk the 3136 tile blocks; the next block's first DMAs are issued "
                                       "from inside the epilogue, one statistics reduction per block) of " if os.environ.get("SLIC_WINO2_PERSIST", "1") != "0" else "") +
                                      "conv_wino2_kernel (64->64 3x3x3 convolution as Winograd F(4,3) x F(2,3) over (W, H): 24 fp32-MFMA GEMMs per kt "
                                      "over tiles of 2 x 4 outputs, H-points split over the waves, LDS-DMA rings of 8-channel pixel stages and 4-channel U stages; forward launches of layer1)") if wino2 else
                                     "conv_wino_kernel<3,false,2> (64->64 3x3x3 convolution as Winograd F(4,3) along W: six fp32-MFMA GEMMs per (kt, kh) "
                                     "over W-tiles, LDS-DMA 3-stage ring, transforms in registers; fwd + dgrad of layer1)" if wino else
                                     "conv_gemm_dma_kernel<128,64,2,2,2,32> (64->64 3x3x3 gather-GEMM, LDS-DMA 2-stage ring; fwd + dgrad of layer1)"),
                             ms_per_launch=ms_k, launches_timed=len(ev),
                             # the same launches in the committed rocprofv3 kernel trace (profiles/): its intervals tile the queue's timeline
                             # (a kernel's interval opens when its predecessor closes, so it carries the dispatch latency between the two),
                             # which reads ~4 % longer than the HIP events around the launch; the fraction at that duration beside `frac`
                             ms_per_launch_rocprof_trace_from_profile=trace_ms_prof,
                             frac_at_rocprof_trace_duration=(executed / (trace_ms_prof * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS
                                                             if (trace_ms_prof and wino2) else None),
                             launches="the four forward launches of layer1 per step (fused BatchNorm statistics); its four data-gradient "
                                      "launches run beside the side stream's weight gradients",
                             ms_per_launch_dgrad_overlapped=float(np.mean(ev_dg)) if ev_dg else None,
                             executed_mfma_flops_per_launch=executed, algorithmic_flops_per_launch=flops_launch,
                             algorithmic_tflops=flops_launch / (ms_k * 1e-3) / 1e12,
                             algorithmic_speedup_vs_direct=flops_launch / executed,
                             note="achieved / frac count the fp32 MFMA FLOPs the kernel executes; algorithmic_* count the direct "
                                  "convolution's 2 M N K over the same time (not a fraction of any hardware peak)"),
               whole_step=dict(executed_gflop_per_clip=exe_step / B / 1e9, algorithmic_gflop_per_clip=alg_step / B / 1e9,
                               executed_tflops=exe_step / (dt / args.steps) / 1e12,
                               frac_of_fp32_mfma_peak_executed=exe_step / (dt / args.steps) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                               algorithmic_tflops=alg_step / (dt / args.steps) / 1e12,
                               note="per GPU; executed = the convolutions' MFMA FLOPs as run (Winograd layers count a half or a third), "
                                    "algorithmic = the direct form's (SURVEY §8a: 248.9 GFLOP per clip)"))
    def all_ranks_ok(ok):
        """every rank reports; False anywhere -> False everywhere (the rows after this point run collectives on all ranks)"""
        if not use_dist:
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
        return bool(t.item())

    # The ONE JSON line of the contract is printed at the very end — but a multi-rank run must never lose its headline to a secondary row
    # that hangs in a collective (a rank that failed alone) or outlives the driver's patience: every rank arms a watchdog with the same
    # budget; when it fires, rank 0 prints the line with what has been measured so far and every rank leaves with exit code 0.
    # Progress also goes to stderr: the headline before the secondary rows start, and again with the k-means rows.
    import threading
    printed = threading.Lock()
    state = dict(done=False)

    headline_only = json.dumps(dict(res, watchdog="secondary rows cut short by the watchdog: headline only"))

    def emit(final):
        with printed:
            if state["done"]:
                return
            state["done"] = True
            if rank == 0:
                try:
                    line = json.dumps(res if final else dict(res, watchdog=f"secondary rows exceeded {budget} s: line printed by the watchdog"))
                except Exception:                 # the main thread was adding a row while the watchdog serialised the dictionary
                    line = headline_only
                print(line, flush=True)

    budget = int(os.environ.get("SLIC_BENCH_SECONDARY_BUDGET_S", "420"))
    wd = None
    if use_dist and world > 1 and not args.no_secondary:
        def fire():
            emit(False)
            log(f"bench.py: rank {rank}: watchdog after {budget} s in the secondary rows; leaving")
            # The headline (measured before any secondary row) is complete and printed: the exit code stays 0 by default so that a launcher does not
            # discard it, and the cut is visible in the line itself ("watchdog" key; tests/test_bench_contract.py treats that key as a failure of
            # the secondary rows).  SLIC_BENCH_WATCHDOG_RC=<n> makes every rank leave with code n instead (drivers that prefer a loud failure).
            os._exit(int(os.environ.get("SLIC_BENCH_WATCHDOG_RC", "0")))
        wd = threading.Timer(budget, fire)
        wd.daemon = True
        wd.start()
    if rank == 0 and use_dist and not args.no_secondary:
        print(json.dumps(dict(res, note="headline only: printed before the secondary rows, which run collectives on all ranks")),
              file=sys.stderr, flush=True)
    if not args.no_secondary:
        sec_ok = True
        try:
            res["secondary"] = kmeans_secondary(rank, world, pg, run_cpu=(rank == 0 and world == 1 and not args.no_cpu_baseline))
        except Exception as e:                                    # never lose the headline line
            res["secondary"] = dict(error=repr(e))
            sec_ok = False
        if rank == 0 and use_dist:
            print(json.dumps(dict(res, note="headline + k-means rows (strong: secondary, weak: secondary.weak_scaled); more rows follow")),
                  file=sys.stderr, flush=True)
        run_cpu_rows = rank == 0 and world == 1 and not args.no_cpu_baseline
        if not args.quick and rank == 0:
            for key, fn in (("retrieval", lambda: retrieval_secondary(run_cpu_rows)), ("nce", nce_secondary)):
                try:
                    res["secondary"][key] = fn()
                except Exception as e:
                    res["secondary"][key] = dict(error=repr(e))
        try:
            # the rows below run DistributedDataParallel steps: every rank joins them, or none does
            if not all_ranks_ok(sec_ok):
                res["secondary"]["extra_error"] = "skipped: a secondary row failed on some rank"
                raise _Skip()
            if args.quick:
                raise _Skip()
            # embedding extraction (evaluate.py:146-205): eval-mode forward only, BN folded into the conv epilogue
            fwd_exe_gflop = fwd_flops(eng, 1)[1] / 1e9
            net.eval()
            with torch.no_grad():
                for _ in range(2):
                    net(x)
                torch.cuda.synchronize()
                t1 = time.time()
                for _ in range(5):
                    net(x)
                torch.cuda.synchronize()
                dte = (time.time() - t1) / 5
            net.train()
            res["secondary"]["extract"] = dict(metric="clips/sec R3D-18 eval-mode forward (embedding extraction), per GPU", value=B / dte,
                                               unit="clips/s", ms_per_batch=dte * 1e3,
                                               frac_of_fp32_mfma_peak_executed=B / dte * fwd_exe_gflop / 1e3 / FP32_MFMA_PEAK_TFLOPS,
                                               algorithmic_tflops=B / dte * 85.17 / 1e3, executed_gflop_per_clip=fwd_exe_gflop)
            # the yaml input size of the shipped configs, 128 x 128 ("also report S = 128"): same step, 1.306x FLOPs
            x128 = torch.from_numpy(np.random.default_rng(11 + rank).standard_normal((B, 3, 16, 128, 128)).astype(np.float32)).cuda()
            xs = x
            try:
                x = x128                                    # step() closes over x
                for _ in range(2):
                    step()
                torch.cuda.synchronize()
                t1 = time.time()
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                dt128 = (time.time() - t1) / 3
                alg128, exe128 = step_flops(net._engine(x128), B)
            finally:
                x = xs
            res["secondary"]["train_128"] = dict(metric="clips/sec R3D-18+NT-Xent training step at 3x16x128x128, per GPU", value=B / dt128,
                                                 unit="clips/s", ms_per_step=dt128 * 1e3,
                                                 frac_of_fp32_mfma_peak_executed=exe128 / dt128 / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                                 algorithmic_tflops=alg128 / dt128 / 1e12)
            del x128
            # north_star's wording ("synthetic 8 x 3 x 16 x 112 x 112 batches") and the shipped configs' own step (BASELINE configs[3],
            # online_train.py:255-392: 13 anchors + 13 positives + 13 second anchors through ONE forward, random_semi_hard mining + the LLC
            # margin term): small batches run the few-tile launch rules (K splits, one-dimensional Winograd where the 2-D form has too
            # few workgroups) that the B = 32 headline never reaches
            from video_similarity_search_amd.loss.triplet_loss import margin_cosine_loss

            def timed_steps(fn, n_warm, n):
                for _ in range(n_warm):
                    fn()
                torch.cuda.synchronize()
                t1 = time.time()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                return (time.time() - t1) / n

            x8 = torch.from_numpy(np.random.default_rng(13 + rank).standard_normal((8, 3, 16, 112, 112)).astype(np.float32)).cuda()
            lab8 = torch.arange(4).repeat(2).cuda()

            def step8():
                emb = model(x8)
                l8, _ = crit(emb, lab8, sampling_strategy='noise_contrastive')
                opt.zero_grad(set_to_none=True)
                l8.backward()
                opt.step()
            dt8 = timed_steps(step8, 3, 10)
            alg8, exe8 = step_flops(net._engine(x8), 8)
            res["secondary"]["train_b8"] = dict(metric="clips/sec R3D-18+NT-Xent training step at B = 8 x 3x16x112x112, per GPU", value=8 / dt8,
                                                unit="clips/s", ms_per_step=dt8 * 1e3,
                                                frac_of_fp32_mfma_peak_executed=exe8 / dt8 / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                                algorithmic_tflops=alg8 / dt8 / 1e12)
            del x8
            x39 = torch.from_numpy(np.random.default_rng(17 + rank).standard_normal((39, 3, 16, 112, 112)).astype(np.float32)).cuda()
            lab26 = torch.arange(13).repeat(2).cuda()

            def step39():
                out = model(x39)
                l39, _ = crit(out[:26], lab26, sampling_strategy="random_semi_hard")
                l39 = l39 + 1.0 * margin_cosine_loss(out[:13], out[26:39], out[13:26], 0.04)
                opt.zero_grad(set_to_none=True)
                l39.backward()
                opt.step()
            dt39 = timed_steps(step39, 2, 5)
            alg39, exe39 = step_flops(net._engine(x39), 39)
            res["secondary"]["train_configs3_b39"] = dict(
                metric="clips/sec R3D-18 training step of BASELINE configs[3]: 39 clips (13 anchors + 13 positives + 13 second anchors), "
                       "random_semi_hard + LLC margin term, per GPU", value=39 / dt39, unit="clips/s", ms_per_step=dt39 * 1e3,
                frac_of_fp32_mfma_peak_executed=exe39 / dt39 / 1e12 / FP32_MFMA_PEAK_TFLOPS, algorithmic_tflops=alg39 / dt39 / 1e12)
            del x39
            if world == 1:
                # the reference-shaped clustering call end to end: KMeans(n_clusters=500, n_init=10) = 10 x (k-means++ + Lloyd, tol 1e-4)
                from video_similarity_search_amd.clustering import fit_cluster
                import contextlib, io
                rngc = np.random.default_rng(1)
                cent = rngc.standard_normal((500, 512)); cent /= np.linalg.norm(cent, axis=1, keepdims=True)
                Xc = torch.from_numpy((cent[rngc.integers(0, 500, 100000)] + 0.35 * rngc.standard_normal((100000, 512)) / np.sqrt(512)).astype(np.float32)).cuda()
                np.random.seed(1)
                with contextlib.redirect_stdout(io.StringIO()):
                    fit_cluster(Xc, 'kmeans', k=500, l2normalize=True)      # first call of the process: code objects load lazily (0.2-0.3 s)
                torch.cuda.synchronize()
                t1 = time.time()
                with contextlib.redirect_stdout(io.StringIO()):
                    fit_cluster(Xc, 'kmeans', k=500, l2normalize=True)
                torch.cuda.synchronize()
                res["secondary"]["fit_cluster_reference_call_seconds"] = time.time() - t1
        except _Skip:
            pass
        except Exception as e:
            res["secondary"]["extra_error"] = repr(e)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_encoder(sd)
    if rank == 0 and world == 1 and not use_dist and not args.no_traffic and not args.no_cpu_baseline and wino2 and os.environ.get("SLIC_BENCH_CHILD") != "1":
        # roofline.traffic LIVE: HBM bytes per launch of the dominant kernel from the PMC counters of THIS
        # command — two child runs of a short bench under rocprofv3 (FETCH_SIZE, WRITE_SIZE: separate --pmc passes with --kernel-trace
        # only, as MI355X_MICROARCH.md prescribes; children, because the counters need the profiler around the process)
        try:
            tr = measure_traffic()
            res["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
            res["roofline"]["traffic_detail"] = tr
        except Exception as e:                                    # the profiler is missing / refused: the committed profile's value stays beside null
            res["roofline"]["traffic_error"] = repr(e)[:300]
    if wd is not None:
        wd.cancel()
    emit(True)
    if use_dist and not args.no_secondary and os.environ.get("SLIC_BENCH_ONESHOT", "1" if world == 1 else "0") != "0":
        # AFTER the contract's line is out: the sharded k-means row once more with the iteration's exchange as the library's one-shot
        # all-to-all over peer-mapped memory (exchange='oneshot', csrc/oneshot.hip) instead of the RCCL all-reduce.  Its waits are bounded
        # (20 s here); the result goes to stderr as its own JSON line.  On by default for the one-rank path test; with more ranks it runs
        # only under SLIC_BENCH_ONESHOT=1 — the exchange between GPUs has not run on hardware yet (one-GPU boxes: two processes on one
        # device is what the tests cover), and a fault there would turn this job's exit code non-zero after a good line.
        try:
            os.environ["SLIC_COMM_TIMEOUT_MS"] = os.environ.get("SLIC_BENCH_ONESHOT_TIMEOUT_MS", "20000")
            os.environ.setdefault("SLIC_ONESHOT_MULTI_GPU", "1")      # asking for this row IS the opt-in to the experimental inter-GPU exchange
            row = kmeans_oneshot_row(rank, world, pg)
            if rank == 0:
                print(json.dumps(dict(row="kmeans_oneshot", **row)), file=sys.stderr, flush=True)
        except Exception as e:
            log(f"bench.py: rank {rank}: one-shot k-means row failed: {e!r}")
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
