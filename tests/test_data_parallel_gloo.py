"""CPU, world_size 2 over gloo: the CONTROL FLOW of misc.distributed_helper.data_parallel with more than one rank — BatchNorm buffers re-pointed
into flat tensors, ignored by DistributedDataParallel and broadcast from rank 0 by the forward pre-hook; the bucket views handed to a model that
writes its gradients into them.  A small torch model stands in for the encoder (the HIP engine needs a GPU: tests/test_dist_gpu.py runs the
same wrapper around it on a one-rank RCCL group; ReduceOp.AVG is RCCL's — over gloo the wrapper keeps torch's default scale, which is what
this test compares with): three training steps under data_parallel == the same steps under the plain wrapper, bit for bit, on both ranks."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(11)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(),
                               torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(),
                               torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(), torch.nn.Linear(8, 4))


def _run(model_ddp, rank, steps=3):
    opt = torch.optim.SGD(model_ddp.parameters(), lr=0.1, momentum=0.5)
    out = []
    for s in range(steps):
        g = torch.Generator().manual_seed(100 * s + rank)            # every rank its own data
        x = torch.randn(6, 3, 8, 8, generator=g)
        loss = model_ddp(x).square().mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        out.append(float(loss.item()))
    return out


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from video_similarity_search_amd.misc.distributed_helper import data_parallel
    torch.distributed.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.set_num_threads(1)
    ref = _model()
    if rank == 1:                                                    # rank 1 starts from DIFFERENT running statistics: rank 0's must win
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.add_(1.0)
    fast = _model()
    fast.load_state_dict(ref.state_dict())
    keys = list(fast.state_dict().keys())
    plain = torch.nn.parallel.DistributedDataParallel(ref)
    l_plain = _run(plain, rank)
    wrapped = data_parallel(fast)
    info = wrapped.slic_ddp
    assert info["flat_buffers"] and info["buffers_flattened"] == 6 and info["buffer_broadcasts_per_forward"] == 2
    assert list(fast.state_dict().keys()) == keys
    l_fast = _run(wrapped, rank)
    same_w = all(torch.equal(a, b) for a, b in zip(ref.parameters(), fast.parameters()))
    same_b = all(torch.equal(a, b) for a, b in zip(ref.buffers(), fast.buffers()))
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), l_plain=np.array(l_plain), l_fast=np.array(l_fast), same_w=same_w, same_b=same_b,
             w0=next(fast.parameters()).detach().numpy(), rm=fast[1].running_mean.numpy())
    torch.distributed.destroy_process_group()


def test_data_parallel_two_ranks_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [dict(np.load(os.path.join(tmp_path, f"r{r}.npz"))) for r in range(world)]
    for r in res:
        assert np.array_equal(r["l_plain"], r["l_fast"])              # the same losses step by step
        assert bool(r["same_w"]) and bool(r["same_b"])                 # the same weights and running statistics as under the plain wrapper
    assert np.array_equal(res[0]["w0"], res[1]["w0"])                  # replicas identical
