"""60 training steps: allocated / reserved HBM must be flat after warm-up (no leak through the engine's caches or side channel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from video_similarity_search_amd.loss import OnlineTripletLoss
model, _ = bench.build_model(); model = model.cuda().train()
opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.5)
crit = OnlineTripletLoss(0.2, 'cosine')
x = torch.randn(32, 3, 16, 112, 112, device="cuda"); lab = torch.arange(16).repeat(2).cuda()
for it in range(60):
    loss, _ = crit(model(x), lab, sampling_strategy='noise_contrastive')
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    if it in (5, 20, 59):
        torch.cuda.synchronize()
        print(it, f"allocated {torch.cuda.memory_allocated()/2**30:.3f} GiB  reserved {torch.cuda.memory_reserved()/2**30:.3f} GiB  peak {torch.cuda.max_memory_allocated()/2**30:.3f} GiB  loss {loss.item():.4f}", flush=True)
eng = next(iter(model._engines.values()))
print("live entries", len(eng._live), "prefused", len(eng._prefused))
