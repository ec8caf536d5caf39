"""GPU-box micro-benchmarks of the other SURVEY.md §8 rows at their BASELINE sizes (one JSON line each):
A8 retrieval 10k x 100k x 512 top-50, A4 memory-bank NCE (B=32, K=1024, D=128, n_data=100k), A3 NT-Xent (2B = 32, D = 128)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch

from video_similarity_search_amd.evaluate import cosine_topk
from video_similarity_search_amd.loss import OnlineTripletLoss
from video_similarity_search_amd.loss.NCE_loss import NCEAverage, NCESoftmaxLoss

PEAK, HBM = 157.3, 8.0


def timeit(f, reps):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


rng = np.random.default_rng(5)
Q = torch.from_numpy(rng.standard_normal((10000, 512)).astype(np.float32)).cuda()
G = torch.from_numpy(rng.standard_normal((100000, 512)).astype(np.float32)).cuda()
t = timeit(lambda: cosine_topk(Q, G, k=50), 3)
fl = 2.0 * 10000 * 100000 * 512
print(json.dumps(dict(row="A8 retrieval", workload="10k x 512 queries vs 100k x 512 gallery, cosine top-50 (normalise + fused top-k + merge)",
                      seconds=t, queries_per_s=10000 / t, tflops=fl / t / 1e12, frac_fp32_mfma=fl / t / 1e12 / PEAK)))

B, K, D, n_data = 32, 1024, 128, 100000
nce = NCEAverage(D, n_data, K).cuda()
crit = NCESoftmaxLoss().cuda()
l = torch.randn(B, D, device="cuda", requires_grad=True)
ab = torch.randn(B, D, device="cuda", requires_grad=True)
y = torch.randint(0, n_data, (B,), device="cuda")


def nce_step():
    o1, o2 = nce(l, ab, y)
    (crit(o1) + crit(o2)).backward()


def nce_step_fused():                  # what contrastive_train_epoch runs: NCEAverage.softmax_loss, three launches + the index draw
    nce.softmax_loss(l, ab, y)[0].backward()


idx_fixed = torch.randint(0, n_data, (B, K + 1), device="cuda")
idx_fixed[:, 0] = y


def nce_step_fused_fixed_idx():        # the same with the negatives given (no torch.randint launch in the timed region)
    nce.softmax_loss(l, ab, y, idx_fixed)[0].backward()


t_mod = timeit(nce_step, 20)
t = timeit(nce_step_fused, 20)
t_fix = timeit(nce_step_fused_fixed_idx, 20)
bytes_alg = 2 * B * (K + 1) * D * 4 + 2 * B * D * 4 * 2
print(json.dumps(dict(row="A4 memory-bank NCE", workload=f"NCEAverage.softmax_loss (scores + cross-entropy, bank update, backward) fwd+bwd, B={B} K={K} D={D} n_data={n_data}",
                      seconds=t, seconds_with_given_negatives=t_fix, seconds_module_by_module=t_mod, algorithmic_bytes=bytes_alg,
                      gb_per_s=bytes_alg / t / 1e9, frac_hbm_8TBs=bytes_alg / t / 1e12 / HBM,
                      note="three launches + the index draw + autograd bookkeeping; the gathers themselves are 33.6 MB")))

emb = torch.randn(32, 128, device="cuda", requires_grad=True)
lab = torch.arange(16).repeat(2).cuda()
crit3 = OnlineTripletLoss(0.2, 'cosine')


def ntx():
    loss, _ = crit3(emb, lab, sampling_strategy='noise_contrastive')
    loss.backward()


t = timeit(ntx, 50)
print(json.dumps(dict(row="A3 NT-Xent", workload="OnlineTripletLoss(noise_contrastive) fwd+bwd on [32, 128]", seconds=t,
                      note="launch-bound (reference: ~2B + 5 launches + host loops; here 4 kernels)")))
