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
