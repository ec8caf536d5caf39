"""diagnostic (GPU box): is a tiny-encoder training step bit-repeatable?  prints the tensors whose gradients differ between two runs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from dist_gpu_worker import tiny_state_dict
from video_similarity_search_amd.loss import OnlineTripletLoss
m, sd0 = tiny_state_dict()
m = m.cuda().train()
x = torch.from_numpy(np.random.default_rng(100).standard_normal((4, 3, 8, 32, 32)).astype(np.float32)).cuda()
labels = torch.arange(2).repeat(2).cuda()
crit = OnlineTripletLoss(0.2, 'cosine')
runs = []
for it in range(3):
    m.load_state_dict(sd0)
    m.zero_grad(set_to_none=True)
    emb = m(x)
    loss, _ = crit(emb, labels, sampling_strategy='noise_contrastive')
    loss.backward()
    torch.cuda.synchronize()
    runs.append((float(loss.item()), emb.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
print("env", {k: v for k, v in os.environ.items() if k.startswith("SLIC_")}, "losses", [r[0] for r in runs])
for it in (1, 2):
    bad = [(k, float((runs[it][2][k] - runs[0][2][k]).abs().max())) for k in runs[0][2] if not torch.equal(runs[it][2][k], runs[0][2][k])]
    print(f"run {it} vs 0: emb equal {torch.equal(runs[it][1], runs[0][1])}; {len(bad)} gradient tensors differ", bad[:8])
