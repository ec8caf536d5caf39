"""round 6: which torch ops launch the copies / fills a training step gains under DistributedDataParallel at one rank (scripts/r6/prof_ddp.sh shows
+75 __amd_rocclr_copyBuffer and +65 fillBufferAligned per step): one step under torch.profiler with Python stacks, aten::copy_ / zero_ / fill_ by caller."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch
import bench
torch.cuda.set_device(0)
torch.distributed.init_process_group(backend="nccl")
from video_similarity_search_amd.loss import OnlineTripletLoss
from video_similarity_search_amd.misc.distributed_helper import data_parallel
model, _ = bench.build_model()
model = model.cuda().train()
fast = os.environ.get("SLIC_DDP_FAST", "1") != "0"
ddp = data_parallel(model, 0) if fast else torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], gradient_as_bucket_view=True)
crit = OnlineTripletLoss(0.2, 'cosine')
opt = torch.optim.SGD(ddp.parameters(), lr=0.1, momentum=0.5)
B = 32
x = torch.randn(B, 3, 16, 112, 112, device="cuda")
labels = torch.arange(B // 2).repeat(2).cuda()
def step():
    emb = ddp(x)
    loss, _ = crit(emb, labels, sampling_strategy='noise_contrastive')
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
for _ in range(4):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_stack_n=6):
    if ev.key in ("aten::copy_", "aten::zero_", "aten::fill_", "aten::zeros", "aten::cat", "aten::mul_", "aten::div_", "aten::mul", "aten::_foreach_add_", "aten::empty_like"):
        st = [s_ for s_ in (ev.stack or []) if "torch/profiler" not in s_][:5]
        rows.append((ev.count, ev.key, " <- ".join(s_.split("/")[-1][:70] for s_ in st) if st else "(no python frame: called from C++ — the DDP reducer / autograd engine)"))
for n, name, where in sorted(rows, reverse=True)[:30]:
    print(f"{n:4d} x {name:14s} {where}")
