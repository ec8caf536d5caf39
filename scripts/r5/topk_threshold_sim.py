"""Candidate counts of the retrieval collect path (csrc/topk.hip: topk_collect_plan), simulated: the scores of a query against the
gallery are i.i.d. from ANY continuous distribution, so only ranks matter — tau = the M-th best of the pooled sample of ns rows sits
at quantile Beta(ns - M + 1, M); every other gallery row passes with probability 1 - tau.
Prints, per (Nq, Ng, k): the plan, the mean candidate count and the two failure probabilities (fewer than k: the query is redone by the
streaming kernels; more than the 2048 slots: likewise).      python scripts/r5/topk_threshold_sim.py"""
import numpy as np

CAP, BG = 2048, 128


def plan(Nq, Ng, k):
    """mirror of topk_collect_plan"""
    if Ng < 32768:
        return False, 0, 0, 0
    qb = -(-Nq // 128)
    target = 6.0 * k + 100.0
    M = min(12, max(4, int(target * 0.04 + 0.5)))
    ns = M * Ng / target
    S1 = min(8, max(1, 256 // qb))
    if ns / S1 < 2 * BG:
        S1 = max(1, int(ns // (2 * BG)))
    per1 = max(2 * BG, -(-int(ns / S1) // BG) * BG)
    M = min(16, max(4, int(target * S1 * per1 / Ng + 0.5)))
    ok = S1 * per1 * 8 <= Ng
    return ok, S1, per1, M


def sim(Ng, S1, per1, M, trials=400000, seed=0):
    rng = np.random.default_rng(seed)
    ns = S1 * per1
    tau = rng.beta(ns - M + 1, M, size=trials)
    return rng.binomial(Ng - ns, 1.0 - tau) + M


if __name__ == "__main__":
    for Nq, Ng, k in [(10000, 100000, 50), (10000, 100000, 1), (10000, 100000, 88), (10000, 100000, 20), (300, 60000, 50), (100, 32768, 50),
                      (1000, 1000000, 50), (33000, 33000, 5), (128, 50000, 88), (5000, 200000, 10)]:
        ok, S1, per1, m1 = plan(Nq, Ng, k)
        if not ok:
            print(f"Nq={Nq} Ng={Ng} k={k}: streaming path")
            continue
        c = sim(Ng, S1, per1, m1)
        print(f"Nq={Nq:6d} Ng={Ng:8d} k={k:3d}: S1={S1} per1={per1:5d} M={m1:2d} sample={S1 * per1 / Ng:5.1%}  candidates mean {c.mean():6.0f} "
              f"p0.1 {np.percentile(c, 0.1):5.0f} p99.9 {np.percentile(c, 99.9):6.0f}  P(<k) {np.mean(c < k):.1e}  P(>cap) {np.mean(c > CAP):.1e}")
