"""GPU parity: fused cosine top-k retrieval and the memory-bank NCE path vs goldens / oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_topk_retrieval_matches_sklearn_golden(gpu, golden_dir):
    from video_similarity_search_amd.evaluate import topk_retrieval, cosine_topk
    g = dict(np.load(os.path.join(golden_dir, "retrieval.npz")))
    hits = topk_retrieval(X_train=g["X_train"], y_train=g["y_train"], X_test=g["X_test"], y_test=g["y_test"], ks=list(g["ks"]))
    assert [hits[int(k)] for k in g["ks"]] == list(g["topk_correct"])            # hit counts identical
    idx, dist = cosine_topk(g["X_test"], g["X_train"], k=50)
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    same = idx == g["top50"]
    # index sets identical except at near-ties of fp32 vs the golden's float64 distances: report the gap
    bad = np.argwhere(~same)
    for q, j in bad[:50]:
        assert abs(dist[q, j] - g["d50"][q, j]) < 2e-6, (q, j, dist[q, j], g["d50"][q, j])
    assert same.mean() > 0.995
    np.testing.assert_allclose(dist, g["d50"], atol=2e-6)
    assert (np.diff(dist, axis=1) >= 0).all()                                    # sorted ascending


def test_topk_retrieval_feature_files(gpu, tmp_path):
    """the on-disk format of iic_retrieve_clips.py:233-236,275-314: {train,test}_feature.npy [V, 10, D] (ten clips per video,
    averaged), {train,test}_class.npy [V, 10]; result also written to topk_correct.json next to them"""
    import json
    from types import SimpleNamespace
    from oracle import retrieval as orr
    from video_similarity_search_amd.evaluate import topk_retrieval
    rng = np.random.default_rng(12)
    Vt, Vq, D = 300, 90, 64
    cls_t, cls_q = rng.integers(0, 11, Vt), rng.integers(0, 11, Vq)
    cent = rng.standard_normal((11, D))
    ft = (cent[cls_t][:, None, :] + 1.5 * rng.standard_normal((Vt, 10, D))).astype(np.float32)
    fq = (cent[cls_q][:, None, :] + 1.5 * rng.standard_normal((Vq, 10, D))).astype(np.float32)
    np.save(tmp_path / "train_feature.npy", ft); np.save(tmp_path / "train_class.npy", np.repeat(cls_t[:, None], 10, 1))
    np.save(tmp_path / "test_feature.npy", fq); np.save(tmp_path / "test_class.npy", np.repeat(cls_q[:, None], 10, 1))
    hits = topk_retrieval(SimpleNamespace(feature_dir=str(tmp_path)))
    ref, _ = orr.topk_retrieval(ft.mean(1), cls_t, fq.mean(1), cls_q, ks=[1, 5, 10, 20, 50])
    assert hits == {int(k): int(v) for k, v in ref.items()}
    assert json.load(open(tmp_path / "topk_correct.json")) == {str(k): v for k, v in hits.items()}


def test_self_retrieval_and_distance_matrix(gpu, golden_dir):
    from video_similarity_search_amd.evaluate import (get_distance_matrix, get_topk_acc, get_topk_acc_from_embeddings,
                                                      cosine_topk)
    g = dict(np.load(os.path.join(golden_dir, "retrieval.npz")))
    X = g["X_train"][:500].astype(np.float32)
    y = g["y_train"][:500]
    dm = get_distance_matrix(X)                       # diag = inf, like evaluate.py:221-222
    assert np.isinf(np.diag(dm)).all()
    off = ~np.eye(500, dtype=bool)
    np.testing.assert_allclose(dm[off], g["self_dm"][off], atol=3e-6)
    acc = get_topk_acc(dm, y)
    np.testing.assert_allclose(acc, g["self_acc"], atol=1e-12)
    acc2 = get_topk_acc_from_embeddings(X, y)
    np.testing.assert_allclose(acc2, g["self_acc"], atol=1e-12)
    idx, _ = cosine_topk(X, None, k=20)
    assert (idx.cpu().numpy() == g["self_top20"]).mean() > 0.995
    assert (idx.cpu().numpy() != np.arange(500)[:, None]).all()                  # never returns itself
    # euclidean branch and a rectangular cosine matrix
    from sklearn.metrics.pairwise import euclidean_distances, cosine_distances
    Y = g["X_test"][:77].astype(np.float32)
    np.testing.assert_allclose(get_distance_matrix(X, Y, 'euclidean'), euclidean_distances(X, Y), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(get_distance_matrix(X, Y, 'cosine'), cosine_distances(X, Y), atol=3e-6)


# every partial kernel of csrc/topk.hip: query operand in registers with 4 / 8 / 16 k-tiles (D <= 128 / 256 / 512), the
# 2-stage ring for D > 512, the short pending columns of a large k, several gallery slices with a ragged last one
@pytest.mark.parametrize("Nq,Ng,D,k", [(1, 50, 8, 50), (129, 1000, 128, 1), (1000, 20000, 512, 20), (33, 257, 40, 7),
                                       (200, 3001, 256, 50), (300, 40000, 200, 50), (64, 2000, 640, 10),
                                       (100, 5000, 512, 88), (130, 700, 1024, 3)])
def test_topk_shapes_vs_oracle(gpu, Nq, Ng, D, k):
    from oracle import retrieval as orr
    from video_similarity_search_amd.evaluate import cosine_topk
    rng = np.random.default_rng(Nq + Ng)
    Q = rng.standard_normal((Nq, D)).astype(np.float32)
    G = rng.standard_normal((Ng, D)).astype(np.float32)
    G[3] = 0.0                                                       # zero row: normalize leaves it, distance 1
    idx, dist = cosine_topk(Q, G, k=k)
    d = orr.cosine_distances(Q.astype(np.float64), G.astype(np.float64))
    ref = np.argsort(d, axis=1, kind="stable")[:, :k]
    refd = np.take_along_axis(d, ref, axis=1)
    np.testing.assert_allclose(dist.cpu().numpy(), refd, atol=3e-6)
    assert (idx.cpu().numpy() == ref).mean() > 0.99


def _topk_plan(Nq, Ng, D, k):
    import ctypes
    from video_similarity_search_amd import _lib
    out = (ctypes.c_int * 6)()
    _lib.check(_lib.load().slic_cosine_topk_plan(Nq, Ng, D, k, out), "slic_cosine_topk_plan")
    return dict(collect=bool(out[0]), S1=out[1], per1=out[2], m1=out[3], gstep=out[4], cap=out[5])


def _topk_env(v):
    import contextlib

    @contextlib.contextmanager
    def cm():
        old = os.environ.get("SLIC_TOPK_COLLECT")
        os.environ["SLIC_TOPK_COLLECT"] = v
        try:
            yield
        finally:
            if old is None:
                del os.environ["SLIC_TOPK_COLLECT"]
            else:
                os.environ["SLIC_TOPK_COLLECT"] = old
    return cm()


def _both_paths(Q, G, k):
    """(idx, dist) of the collect path (SLIC_TOPK_COLLECT=1: for any k) and of the streaming path (=0); the library reads the switch at every call"""
    from video_similarity_search_amd.evaluate import cosine_topk
    with _topk_env("1"):
        a = cosine_topk(Q, G, k=k)
    with _topk_env("0"):
        b = cosine_topk(Q, G, k=k)
    return [t.cpu().numpy() for t in a], [t.cpu().numpy() for t in b]


# the collect path (threshold -> collect -> select, csrc/topk.hip) against the streaming kernels: the same scores (same k order inside
# every accumulator) and the same tie rule, so the lists must be IDENTICAL, bit for bit; and against the float64 oracle
@pytest.mark.parametrize("Nq,Ng,D,k", [(300, 40000, 200, 50), (1000, 100000, 512, 50), (257, 65613, 128, 1), (128, 50000, 512, 88),
                                       (5, 32768, 64, 20), (2500, 33000, 256, 7)])
def test_topk_collect_equals_streaming(gpu, Nq, Ng, D, k):
    from oracle import retrieval as orr
    with _topk_env("1"):
        assert _topk_plan(Nq, Ng, D, k)["collect"]
    assert _topk_plan(Nq, Ng, D, k)["collect"] == (k >= 16)          # the default: where it is faster
    rng = np.random.default_rng(Nq + Ng + k)
    Q = rng.standard_normal((Nq, D)).astype(np.float32)
    G = rng.standard_normal((Ng, D)).astype(np.float32)
    G[3] = 0.0
    (ia, da), (ib, db) = _both_paths(Q, G, k)
    assert np.array_equal(ia, ib) and np.array_equal(da, db)
    sub = rng.choice(Nq, min(Nq, 48), replace=False)
    d = orr.cosine_distances(Q[sub].astype(np.float64), G.astype(np.float64))
    ref = np.argsort(d, axis=1, kind="stable")[:, :k]
    np.testing.assert_allclose(da[sub], np.take_along_axis(d, ref, axis=1), atol=3e-6)
    assert (ia[sub] == ref).mean() > 0.99


def test_topk_collect_fuzz_equals_streaming(gpu):
    """random shapes over the collect path's whole domain — any query count, galleries from 32 768 rows (ragged last tiles and slices),
    every D % 8 == 0 up to 512 (the three register-operand instantiations), k from 1 to 88 (forced on below 16), with and without the
    self mask, score distributions from flat to heavily clustered (many near-ties) — the two algorithms must return identical lists"""
    from video_similarity_search_amd.evaluate import cosine_topk
    rng = np.random.default_rng(2025)
    for trial in range(14):
        Nq = int(rng.integers(1, 700))
        Ng = int(rng.integers(32768, 90000))
        D = int(rng.integers(1, 65)) * 8
        k = int(rng.choice([1, 3, 16, 17, 31, 50, 64, 88]))
        self_mask = trial % 5 == 4
        if self_mask:
            Nq = Ng = int(rng.integers(32768, 40000))
        G = rng.standard_normal((Ng, D)).astype(np.float32)
        if trial % 3 == 1:                                          # clustered gallery: 40 directions + small noise -> crowds of near-equal scores
            cent = rng.standard_normal((40, D)).astype(np.float32)
            G = (cent[rng.integers(0, 40, Ng)] + 0.05 * rng.standard_normal((Ng, D))).astype(np.float32)
        if trial % 3 == 2:                                          # exact duplicates of a few rows sprinkled in (ties broken by index)
            dup = rng.integers(0, Ng, 200)
            G[dup] = G[dup[0]]
        Q = None if self_mask else (G[rng.integers(0, Ng, Nq)] + 0.3 * rng.standard_normal((Nq, D))).astype(np.float32)
        with _topk_env("1"):
            assert _topk_plan(Nq, Ng, D, k)["collect"], (Nq, Ng, D, k)
        (ia, da), (ib, db) = _both_paths(G if self_mask else Q, None if self_mask else G, k)
        assert np.array_equal(ia, ib) and np.array_equal(da, db), (trial, Nq, Ng, D, k, self_mask)


def test_topk_collect_self_mask_equals_streaming(gpu):
    """the self-retrieval form (queries = gallery, the diagonal skipped: evaluate.py:221-222; FINCH's first neighbours)"""
    rng = np.random.default_rng(77)
    X = rng.standard_normal((33000, 128)).astype(np.float32)
    with _topk_env("1"):
        assert _topk_plan(33000, 33000, 128, 5)["collect"]
    (ia, da), (ib, db) = _both_paths(X, None, 5)
    assert np.array_equal(ia, ib) and np.array_equal(da, db)
    assert not np.any(ia == np.arange(33000)[:, None])


def test_topk_collect_fallback_too_many_candidates(gpu):
    """a gallery of 64 distinct rows repeated 4096 times: every score a query reaches is reached by 4096 rows, the candidate buffers
    (2048 slots) overflow, and every query is redone by the streaming kernels — exact, ties to the lower index"""
    from oracle import retrieval as orr
    rng = np.random.default_rng(3)
    base = rng.standard_normal((64, 64)).astype(np.float32)
    G = np.tile(base, (4096, 1))                                   # row i = base[i % 64]
    Q = rng.standard_normal((200, 64)).astype(np.float32)
    k = 50
    pl = _topk_plan(200, G.shape[0], 64, k)
    assert pl["collect"] and pl["cap"] < 4096
    (ia, da), (ib, db) = _both_paths(Q, G, k)
    assert np.array_equal(ia, ib) and np.array_equal(da, db)
    d = orr.cosine_distances(Q.astype(np.float64), base.astype(np.float64))
    best = np.argmin(d, axis=1)                                    # the nearest distinct row; its 4096 copies fill the list, lowest index first
    # (fp32 scores of identical rows are identical, so the order inside the list is by index)
    assert np.array_equal(ia, best[:, None] + 64 * np.arange(k)[None, :])


def test_topk_collect_fallback_too_few_candidates(gpu):
    """an unrepresentative sample: the only gallery rows close to the queries sit exactly where the strided sample looks, so the
    thresholds come out far above what the rest of the gallery reaches, fewer than k rows pass, and the streaming kernels redo
    every query"""
    from oracle import retrieval as orr
    Nq, Ng, D, k = 300, 60000, 128, 50
    pl = _topk_plan(Nq, Ng, D, k)
    assert pl["collect"] and pl["m1"] < k
    rng = np.random.default_rng(9)
    v = rng.standard_normal(D).astype(np.float32)
    Q = (v[None, :] + 0.05 * rng.standard_normal((Nq, D))).astype(np.float32)
    G = rng.standard_normal((Ng, D)).astype(np.float32)
    for j in range(pl["m1"]):                                       # M sample rows (sample row i = gallery row i * gstep) next to every query
        G[(pl["per1"] * (j % pl["S1"]) + 7 * j) * pl["gstep"]] = v + 0.01 * rng.standard_normal(D)
    (ia, da), (ib, db) = _both_paths(Q, G, k)
    assert np.array_equal(ia, ib) and np.array_equal(da, db)
    d = orr.cosine_distances(Q.astype(np.float64), G.astype(np.float64))
    ref = np.argsort(d, axis=1, kind="stable")[:, :k]
    np.testing.assert_allclose(da, np.take_along_axis(d, ref, axis=1), atol=3e-6)
    assert (ia == ref).mean() > 0.99


def test_topk_k_limit(gpu):
    """k <= 88 on both paths; beyond it the call fails loudly (include/slic_hip.h)"""
    from video_similarity_search_amd import _lib
    from video_similarity_search_amd.evaluate import cosine_topk
    rng = np.random.default_rng(1)
    Q = rng.standard_normal((10, 64)).astype(np.float32)
    G = rng.standard_normal((40000, 64)).astype(np.float32)
    idx, _ = cosine_topk(Q, G, k=88)
    assert idx.shape == (10, 88)
    with pytest.raises(_lib.SlicError):
        cosine_topk(Q, G, k=89)
    with pytest.raises(_lib.SlicError):
        cosine_topk(Q, G[:500], k=89)


def test_retrieval_full_size_properties(gpu):
    """BASELINE configs[4] size (10k x 512 queries vs 100k x 512 gallery, k = 50), checked through properties the size does
    not change: per-query distances ascending, indices unique and in range, a random subset of queries equal to the exact
    float64 top-k (ties aside), and hit counts of that subset equal to the oracle's."""
    from oracle import retrieval as orr
    from video_similarity_search_amd.evaluate import cosine_topk
    rng = np.random.default_rng(5)
    Nq, Ng, D, k = 10000, 100000, 512, 50
    Q = rng.standard_normal((Nq, D)).astype(np.float32)
    G = rng.standard_normal((Ng, D)).astype(np.float32)
    idx, dist = cosine_topk(torch.from_numpy(Q).cuda(), torch.from_numpy(G).cuda(), k=k)
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    assert idx.shape == (Nq, k) and dist.shape == (Nq, k)
    assert np.all(np.diff(dist, axis=1) >= 0)
    assert idx.min() >= 0 and idx.max() < Ng
    srt = np.sort(idx, axis=1)
    assert np.all(srt[:, 1:] != srt[:, :-1])                       # no duplicates inside a list
    sub = rng.choice(Nq, 64, replace=False)
    Gn = G.astype(np.float64); Gn /= np.linalg.norm(Gn, axis=1, keepdims=True)
    Qs = Q[sub].astype(np.float64); Qs /= np.linalg.norm(Qs, axis=1, keepdims=True)
    d = np.clip(1.0 - Qs @ Gn.T, 0.0, 2.0)
    ref = np.argsort(d, axis=1, kind="stable")[:, :k]
    for row, qi in enumerate(sub):
        got = idx[qi]
        if not np.array_equal(got, ref[row]):                       # only near-ties (fp32 vs fp64 scores) may swap
            bad = np.flatnonzero(got != ref[row])
            assert np.all(np.abs(d[row, got[bad]] - d[row, ref[row][bad]]) < 5e-6), qi
        np.testing.assert_allclose(dist[qi], d[row, got], atol=5e-6)
    labels_g = rng.integers(0, 101, Ng)
    labels_q = rng.integers(0, 101, Nq)
    for kk in (1, 5, 10, 20, 50):
        hit_gpu = sum(int(labels_q[qi] in labels_g[idx[qi, :kk]]) for qi in sub)
        hit_ref = sum(int(labels_q[qi] in labels_g[ref[row, :kk]]) for row, qi in enumerate(sub))
        assert hit_gpu == hit_ref


def test_nce_average_matches_reference_golden(gpu, golden_dir):
    from video_similarity_search_amd.loss import NCEAverage, NCESoftmaxLoss
    g = dict(np.load(os.path.join(golden_dir, "loss_ntxent.npz")))
    B, D = g["nce_l"].shape
    ndata, K = g["nce_memory_l"].shape[0], g["nce_idx"].shape[1] - 1
    nce = NCEAverage(D, ndata, K, 0.07, 0.5).cuda()
    assert set(dict(nce.named_buffers())) == {"params", "memory_l", "memory_ab"}
    nce.memory_l.copy_(torch.from_numpy(g["nce_memory_l"]))
    nce.memory_ab.copy_(torch.from_numpy(g["nce_memory_ab"]))
    l = torch.from_numpy(g["nce_l"]).cuda().requires_grad_(True)
    ab = torch.from_numpy(g["nce_ab"]).cuda().requires_grad_(True)
    out_l, out_ab = nce(l, ab, torch.from_numpy(g["nce_y"]).cuda(), torch.from_numpy(g["nce_idx"]).cuda())
    assert out_l.shape == (B, K + 1, 1)
    np.testing.assert_allclose(out_l.detach().cpu().numpy(), g["nce_out_l"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(out_ab.detach().cpu().numpy(), g["nce_out_ab"], atol=2e-5, rtol=1e-5)
    crit = NCESoftmaxLoss()
    tot = crit(out_l) + crit(out_ab)
    tot.backward()
    assert abs(tot.item() - float(g["nce_loss"])) < 1e-4
    np.testing.assert_allclose(l.grad.cpu().numpy(), g["nce_grad_l"], atol=1e-5, rtol=1e-3)
    np.testing.assert_allclose(ab.grad.cpu().numpy(), g["nce_grad_ab"], atol=1e-5, rtol=1e-3)
    np.testing.assert_allclose(nce.memory_l.cpu().numpy(), g["nce_memory_l_after"], atol=1e-6)
    np.testing.assert_allclose(nce.memory_ab.cpu().numpy(), g["nce_memory_ab_after"], atol=1e-6)
    # default path: idx drawn on the device, column 0 = y
    o1, o2 = nce(l.detach(), ab.detach(), torch.from_numpy(g["nce_y"]).cuda())
    assert o1.shape == (B, K + 1, 1) and torch.isfinite(o1).all()


@pytest.mark.parametrize("dup", [False, True])
def test_nce_fused_step_equals_separate_modules(gpu, golden_dir, dup):
    """NCEAverage.softmax_loss (three launches) == NCESoftmaxLoss(out_l) + NCESoftmaxLoss(out_ab) of the module-by-module path:
    loss, both feature gradients, both banks after the update — on the reference golden (and with a duplicated label: the last
    row wins in both), and at the BASELINE shape B = 32, K = 1024"""
    from video_similarity_search_amd.loss import NCEAverage, NCESoftmaxLoss
    g = dict(np.load(os.path.join(golden_dir, "loss_ntxent.npz")))
    B, D = g["nce_l"].shape
    ndata, K = g["nce_memory_l"].shape[0], g["nce_idx"].shape[1] - 1
    y = torch.from_numpy(g["nce_y"]).cuda()
    idx = torch.from_numpy(g["nce_idx"]).cuda()
    if dup:
        y = y.clone()
        idx = idx.clone()
        y[-1] = y[0]
        idx[-1, 0] = y[0]
    res = []
    for fused in (False, True):
        nce = NCEAverage(D, ndata, K, 0.07, 0.5).cuda()
        nce.memory_l.copy_(torch.from_numpy(g["nce_memory_l"]))
        nce.memory_ab.copy_(torch.from_numpy(g["nce_memory_ab"]))
        l = torch.from_numpy(g["nce_l"]).cuda().requires_grad_(True)
        ab = torch.from_numpy(g["nce_ab"]).cuda().requires_grad_(True)
        if fused:
            tot, out_l, out_ab = nce.softmax_loss(l, ab, y, idx)
            assert not out_l.requires_grad
        else:
            out_l, out_ab = nce(l, ab, y, idx)
            tot = NCESoftmaxLoss()(out_l) + NCESoftmaxLoss()(out_ab)
        (tot * 1.5).backward()
        res.append((tot.item(), out_l.detach().clone(), out_ab.detach().clone(), l.grad.clone(), ab.grad.clone(),
                    nce.memory_l.clone(), nce.memory_ab.clone()))
    a, b = res
    assert abs(a[0] - b[0]) < 1e-5
    if not dup:
        assert abs(b[0] - float(g["nce_loss"])) < 1e-4
        np.testing.assert_allclose(b[3].cpu().numpy() / 1.5, g["nce_grad_l"], atol=1e-5, rtol=1e-3)
        np.testing.assert_allclose(b[5].cpu().numpy(), g["nce_memory_l_after"], atol=1e-6)
    for i in (1, 2):
        assert torch.equal(a[i], b[i])
    for i in (3, 4):
        assert torch.allclose(a[i], b[i], atol=1e-6, rtol=1e-4), i
    for i in (5, 6):
        if dup:      # the module-by-module update races on a duplicated label (two waves, one row); compare the other rows
            keep = torch.ones(ndata, dtype=torch.bool, device="cuda")
            keep[y[0]] = False
            assert torch.equal(a[i][keep], b[i][keep])
            # the fused update: the LAST row with that label wins, computed from the bank as it was
            which, feat = (g["nce_memory_l"], g["nce_l"]) if i == 5 else (g["nce_memory_ab"], g["nce_ab"])
            v = which[int(y[0])] * 0.5 + feat[-1] * 0.5
            np.testing.assert_allclose(b[i][y[0]].cpu().numpy(), v / np.linalg.norm(v), atol=1e-6)
        else:
            assert torch.equal(a[i], b[i]), i
    # BASELINE shape
    Bb, Kb, Db, n = 32, 1024, 128, 100000
    gen = torch.Generator(device="cuda").manual_seed(1)
    nce = NCEAverage(Db, n, Kb).cuda()
    l = torch.randn(Bb, Db, device="cuda", generator=gen, requires_grad=True)
    ab = torch.randn(Bb, Db, device="cuda", generator=gen, requires_grad=True)
    yb = torch.randperm(n, device="cuda", generator=gen)[:Bb]
    ib = torch.randint(0, n, (Bb, Kb + 1), device="cuda", generator=gen)
    ib[:, 0] = yb
    m0 = nce.memory_l.clone(), nce.memory_ab.clone()
    o_l, o_ab = nce(l, ab, yb, ib)
    t0 = NCESoftmaxLoss()(o_l) + NCESoftmaxLoss()(o_ab)
    t0.backward()
    g0 = l.grad.clone(), ab.grad.clone()
    m1 = nce.memory_l.clone(), nce.memory_ab.clone()
    nce.memory_l.copy_(m0[0]); nce.memory_ab.copy_(m0[1])
    l.grad = ab.grad = None
    t1, _, _ = nce.softmax_loss(l, ab, yb, ib)
    t1.backward()
    assert abs(t0.item() - t1.item()) < 1e-4 * abs(t0.item())
    assert torch.allclose(l.grad, g0[0], atol=1e-6, rtol=1e-3) and torch.allclose(ab.grad, g0[1], atol=1e-6, rtol=1e-3)
    assert torch.equal(nce.memory_l, m1[0]) and torch.equal(nce.memory_ab, m1[1])


def test_nce_full_size_properties(gpu):
    """BASELINE shape B=32, K=1024, D=128: scores vs a torch gather on the same device data (linearity / checksum)"""
    from video_similarity_search_amd._lib import call, ptr, stream
    B, K1, D, n = 32, 1025, 128, 100000
    gen = torch.Generator(device="cuda").manual_seed(0)
    bank = torch.randn(n, D, device="cuda", generator=gen)
    f = torch.randn(B, D, device="cuda", generator=gen)
    idx = torch.randint(0, n, (B, K1), device="cuda", generator=gen)
    out = torch.empty(B, K1, device="cuda")
    rows = torch.empty(B * K1, D, device="cuda")
    call("slic_nce_scores_fwd", ptr(bank), ptr(idx), ptr(f), B, K1, D, 0.07, ptr(out), ptr(rows), stream())
    ref = torch.einsum("bkd,bd->bk", bank[idx], f) / 0.07
    assert torch.allclose(out, ref, atol=1e-3, rtol=1e-5)
    assert torch.equal(rows.view(B, K1, D), bank[idx])


def test_sharded_gallery_merge_single_rank_group(gpu):
    """gallery-sharded retrieval: shards searched separately + slic_topk_merge_lists == unsharded search;
    plus the RCCL path itself on a one-rank group"""
    import os
    import torch.distributed as dist
    from video_similarity_search_amd._lib import call, ptr, stream
    from video_similarity_search_amd.evaluate import cosine_topk, cosine_topk_sharded
    rng = np.random.default_rng(8)
    Q = rng.standard_normal((200, 64)).astype(np.float32)
    G = rng.standard_normal((3000, 64)).astype(np.float32)
    k = 10
    ref_i, ref_d = cosine_topk(Q, G, k=k)
    parts_i, parts_d, off = [], [], 0
    for sh in np.array_split(G, 3):
        i, d = cosine_topk(Q, sh, k=k)
        parts_i.append(i + off)
        parts_d.append(d)
        off += len(sh)
    pi, pd = torch.stack(parts_i).contiguous(), torch.stack(parts_d).contiguous()
    oi, od = torch.empty_like(ref_i), torch.empty_like(ref_d)
    call("slic_topk_merge_lists", ptr(pd), ptr(pi), 3, 200, k, ptr(oi), ptr(od), stream())
    assert torch.equal(oi, ref_i) and torch.equal(od, ref_d)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        si, sd = cosine_topk_sharded(Q, G, k, dist.group.WORLD)
        assert torch.equal(si, ref_i) and torch.equal(sd, ref_d)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tag", ["pairs", "clusters", "easy"])
def test_triplet_mining_fixed_semi_hard_matches_reference(gpu, golden_dir, tag):
    """the deterministic selector: triplet indices, loss, n_triplets and the embedding gradient equal the reference's
    (incl. the hardest-easy fallback that returns a position in the negatives list)"""
    from video_similarity_search_amd.loss.triplet_loss import OnlineTripletLoss, get_triplets
    g = dict(np.load(os.path.join(golden_dir, "loss_ntxent.npz")))
    E, labs, margin = g[f"trip_{tag}_E"], g[f"trip_{tag}_labels"], float(g[f"trip_{tag}_margin"])
    e = torch.from_numpy(E).cuda().requires_grad_(True)
    l = torch.from_numpy(labs).cuda()
    a, p, n = get_triplets(e, l, margin, "fixed_semi_hard")
    ref = g[f"trip_{tag}_idx"]
    assert np.array_equal(a.cpu().numpy(), ref[0]) and np.array_equal(p.cpu().numpy(), ref[1])
    assert np.array_equal(n.cpu().numpy(), ref[2])
    loss, nt = OnlineTripletLoss(margin, 'cosine')(e, l, sampling_strategy='fixed_semi_hard')
    loss.backward()
    assert nt == int(g[f"trip_{tag}_n"])
    assert abs(loss.item() - float(g[f"trip_{tag}_loss"])) < 1e-6
    np.testing.assert_allclose(e.grad.cpu().numpy(), g[f"trip_{tag}_grad"], atol=2e-7, rtol=1e-4)


def test_triplet_mining_random_strategies_properties(gpu, golden_dir):
    """random_semi_hard (the shipped default) / random_negative: every chosen negative is a member of the reference's
    candidate set for its pair, choices vary between calls, and the loss equals the oracle's formula on those triplets"""
    from video_similarity_search_amd.loss.triplet_loss import OnlineTripletLoss, get_triplets, pdist
    g = dict(np.load(os.path.join(golden_dir, "loss_ntxent.npz")))
    E, labs, margin = g["trip_clusters_E"], g["trip_clusters_labels"], 0.2
    e, l = torch.from_numpy(E).cuda(), torch.from_numpy(labs).cuda()
    D = pdist(e, 0, 'cosine').cpu().numpy()
    seen = set()
    for strat in ("random_semi_hard", "random_negative"):
        for rep in range(4):
            a, p, n = [t.cpu().numpy() for t in get_triplets(e, l, margin, strat)]
            assert len(a) == int(g["trip_clusters_n"])
            for ai, pi, ni in zip(a, p, n):
                negs = np.nonzero(labs != labs[ai])[0]
                if strat == "random_negative":
                    assert ni in negs
                else:
                    cand = negs[(D[ai, pi] + margin - D[ai, negs]) > 0]
                    if len(cand):
                        assert ni in cand
                    else:
                        assert ni == int(np.argmin(D[ai, negs]))          # reference quirk: list position
            seen.add(tuple(n.tolist()))
    assert len(seen) > 2                                                     # the RNG is actually used
    er = e.clone().requires_grad_(True)
    torch.manual_seed(3)
    loss, nt = OnlineTripletLoss(margin, 'cosine')(er, l, sampling_strategy='random_semi_hard')
    loss.backward()
    assert nt == int(g["trip_clusters_n"]) and torch.isfinite(er.grad).all() and loss.item() > 0
    # no label with two members -> no triplets -> zero loss, like the reference
    z, nz = OnlineTripletLoss(margin, 'cosine')(e[:5], torch.arange(5).cuda(), sampling_strategy='random_semi_hard')
    assert nz == 0 and float(z) == 0.0
