"""
Drop-in for the reference's misc/distributed_helper.py (the only parallelism strategy of SLIC: one process
per GPU, data parallel): run_process / launch_processes / all_reduce / all_gather / is_master_proc /
get_world_size with the same signatures (misc/distributed_helper.py:8-82).

On MI355X `backend="nccl"` IS RCCL (collectives over xGMI); the string stays.  Differences, all
environment-driven: the rendezvous defaults to 127.0.0.1 (container hostnames may not resolve),
HSA_ENABLE_IPC_MODE_LEGACY=0 is kept in the children's environment (dmabuf IPC), and `dist_backend`
may be "gloo" so the same call graph runs in CPU-only tests.
"""
import os

import torch


def run_process(local_rank_proc, NUM_PROC_PER_SHARD, func, shard_id, NUM_SHARDS, cmd_args, cfg,
                proc_init_method="tcp://127.0.0.1:9999", dist_backend="nccl"):
    WORLD_SIZE = NUM_PROC_PER_SHARD * NUM_SHARDS
    rank_proc = shard_id * NUM_PROC_PER_SHARD + local_rank_proc
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if dist_backend == "nccl":
        torch.cuda.set_device(local_rank_proc)          # one GPU per process
    try:
        torch.distributed.init_process_group(backend=dist_backend, init_method=proc_init_method,
                                             world_size=WORLD_SIZE, rank=rank_proc)
        print('Initialized gpu process:', rank_proc)
    except Exception as e:
        print('Failed due to:{}'.format(e))
        raise e
    func(cmd_args, cfg)


def launch_processes(cmd_args, cfg, func, shard_id, NUM_SHARDS, ip_address_port, dist_backend="nccl"):
    if cfg.NUM_GPUS > 1:
        torch.multiprocessing.spawn(fn=run_process, nprocs=cfg.NUM_GPUS,
                                    args=(cfg.NUM_GPUS, func, shard_id, NUM_SHARDS, cmd_args, cfg, ip_address_port,
                                          dist_backend))
    else:
        func(cmd_args, cfg)


def all_reduce(tensors, avg=True):
    """in-place sum (mean by default) across all processes"""
    for tensor in tensors:
        torch.distributed.all_reduce(tensor)
    if avg:
        world_size = torch.distributed.get_world_size()
        for tensor in tensors:
            tensor.mul_(1.0 / world_size)
    return tensors


def all_gather(tensors):
    """every tensor concatenated over ranks along dim 0 (one collective per tensor, rank order)"""
    world_size = torch.distributed.get_world_size()
    output_tensor = []
    for tensor in tensors:
        t = tensor.contiguous()
        out = torch.empty((world_size * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        torch.distributed.all_gather_into_tensor(out, t)
        output_tensor.append(out)
    return output_tensor


def is_master_proc(num_gpus=None):
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_rank() == 0
    return True


def get_world_size():
    if not torch.distributed.is_available():
        return 1
    if not torch.distributed.is_initialized():
        return 1
    return torch.distributed.get_world_size()


# ------------------------------------------------------------------------------------------------------------------------------
# DistributedDataParallel around the HIP encoder: the reference's call (online_train.py:485-494) keeps working unchanged —
#     torch.nn.parallel.DistributedDataParallel(module=model, device_ids=[device])
# — and costs 4-5 % of a step at ONE rank before a byte crosses a link (round 5, scripts/r5/prof_ddp.sh: 160 small copies — the 63
# BatchNorm buffers concatenated, broadcast and copied back at every forward, 66 gradients copied into their bucket views — 66
# per-parameter divisions, 1.9 ms of gaps on the backward's stream).  data_parallel() is the same wrapper with those removed:
#   * buffers: the BatchNorm running statistics (and their int64 counters) are re-pointed into ONE flat tensor per dtype — the
#     registered buffers, their names and state_dict keys stay — DistributedDataParallel is told to leave them alone, and a forward
#     pre-hook broadcasts the flat tensors from rank 0: two collectives per forward, no flatten / unflatten copies (same semantics
#     as broadcast_buffers=True: rank 0's statistics win before every forward);
#   * gradients: with gradient_as_bucket_view the engine writes every weight gradient STRAIGHT into the parameter's bucket view (the
#     view a parameter's .grad held after the previous step, handed to the engine before each forward): autograd installs it as .grad,
#     DistributedDataParallel finds it aliasing its bucket and copies nothing (a view gone stale — buckets rebuilt, .grad replaced by
#     the user — is just another tensor: the copy happens as before);
#   * the 1 / world scale: one all-reduce with ReduceOp.AVG per bucket as the communication hook instead of a division per parameter.
# ------------------------------------------------------------------------------------------------------------------------------
def _flatten_buffers(module):
    """re-point every floating / int64 buffer of `module` into one flat tensor per dtype (the buffers keep their identity, names and
    values); returns ([flat tensors], [buffer names])"""
    groups = {}
    for name, buf in module.named_buffers():
        if buf is None or buf.dtype not in (torch.float32, torch.int64):
            continue
        groups.setdefault(buf.dtype, []).append((name, buf))
    flats, names = [], []
    for dtype, items in groups.items():
        # 16-byte aligned slots: the BatchNorm kernels read the running statistics with the alignment torch's allocator gave them
        align = 16 // torch.empty((), dtype=dtype).element_size()
        offs, tot = [], 0
        for _, b in items:
            offs.append(tot)
            tot += (max(b.numel(), 1) + align - 1) // align * align
        flat = torch.zeros(tot, dtype=dtype, device=items[0][1].device)
        for (name, b), o in zip(items, offs):
            view = flat[o:o + b.numel()].view(b.shape)
            view.copy_(b)
            b.data = view
            names.append(name)
        flats.append(flat)
    return flats, names


def data_parallel(model, device=None, process_group=None, broadcast_buffers=True, bucket_cap_mb=25, **ddp_kwargs):
    """DistributedDataParallel(module=model, device_ids=[device]) for a model of this package, without the wrapper's per-step copies
    (see above).  Returns the DistributedDataParallel instance; `.slic_ddp` on it says what was set up."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    on_gpu = next(model.parameters()).is_cuda
    if device is None and on_gpu:
        device = torch.cuda.current_device()
    pg = process_group if process_group is not None else torch.distributed.group.WORLD
    info = dict(flat_buffers=False, buffer_broadcasts_per_forward=None, gradient_into_bucket_views=False, scale="per-parameter division (torch default)")
    flats, names = [], []
    if broadcast_buffers:
        flats, names = _flatten_buffers(model)
        if names:
            DDP._set_params_and_buffers_to_ignore_for_model(model, names)
            info.update(flat_buffers=True, buffer_broadcasts_per_forward=len(flats), flat_buffer_bytes=[int(f.numel() * f.element_size()) for f in flats],
                        buffers_flattened=len(names))
        for mod in model.modules():                      # plans cache nothing of the buffers, but an engine built before the re-pointing is dropped anyway
            if hasattr(mod, "_engines"):
                mod._engines = {}
    ddp = DDP(model, device_ids=[device] if on_gpu else None, process_group=pg, broadcast_buffers=broadcast_buffers, gradient_as_bucket_view=True,
              bucket_cap_mb=bucket_cap_mb, **ddp_kwargs)      # (a CPU model over gloo — tests — takes the same route; ReduceOp.AVG is RCCL's: gloo keeps the default scale)
    world = torch.distributed.get_world_size(pg)

    def avg_hook(group, bucket):
        fut = torch.distributed.all_reduce(bucket.buffer(), op=torch.distributed.ReduceOp.AVG, group=group, async_op=True).get_future()
        return fut.then(lambda f: f.value()[0])

    def identity_hook(_group, bucket):
        # a group of ONE rank: the mean over the ranks is the bucket itself — no collective, no kernel (RCCL runs a pre-multiplied copy of
        # the whole bucket for AVG even at one rank: 0.24 ms of kernels per step that also hold compute units the convolutions want)
        fut = torch.futures.Future()
        fut.set_result(bucket.buffer())
        return fut

    if torch.distributed.get_backend(pg) == "nccl":
        ddp.register_comm_hook(pg, identity_hook if world == 1 else avg_hook)
        info["scale"] = ("none needed: one rank (identity hook, no collective)" if world == 1 else
                         "ReduceOp.AVG inside the bucket's all-reduce (communication hook)")
    views = {}
    for mod in model.modules():
        if hasattr(mod, "_engines"):
            mod._slic_grad_views = views
    info["gradient_into_bucket_views"] = True
    params = [p for p in model.parameters() if p.requires_grad]

    def pre_forward(_mod, _args):
        # (1) rank 0's statistics to everyone: what broadcast_buffers=True does, as one collective per dtype and no copies
        if flats and (world > 1 or os.environ.get("SLIC_DDP_BROADCAST_AT_ONE_RANK", "0") != "0"):
            for f in flats:
                torch.distributed.broadcast(f, src=torch.distributed.get_global_rank(pg, 0) if hasattr(torch.distributed, "get_global_rank") else 0,
                                            group=pg)
        # (2) the bucket views the gradients of the previous step live in: the engine writes the next gradients there
        for p in params:
            g = p.grad
            if g is not None and g._is_view() and g.is_contiguous() and g.shape == p.shape:
                views[p] = g
        return None

    ddp.register_forward_pre_hook(pre_forward)
    ddp.slic_ddp = info
    return ddp
