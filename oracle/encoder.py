"""
ORACLE — TEST INFRASTRUCTURE ONLY (not the product path; only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this).

CPU restatement (PyTorch CPU ops, fp32 or fp64) of the reference's encoder and InfoNCE loss, written
functionally over a plain state_dict so that it needs nothing from /root/reference at run time:

  encoder_forward      <- models/resnet.py:255-312 (ResNet.forward), :41-57 (BasicBlock.forward),
                          :223-233 (shortcut 'B' = conv1x1x1 + BN), :294-299 (projection head)
  ntxent_loss          <- loss/triplet_loss.py:95-116 ('noise_contrastive') + pdist :429-437
  nce_average / nce_softmax_loss <- loss/NCE_loss.py:26-88, 341-352
  sgd_step             <- online_train.py:543 (SGD lr .1, momentum .5, no weight decay)
  r3d_to_resnet_keys   <- models/r3d/r3d.py:126-187 (R3DNet = the same network under other key names)

Pinned against the imported reference modules by tests/golden/make_goldens_encoder.py (goldens in
tests/golden/encoder_tiny.npz, loss_ntxent.npz), checked by tests/test_oracle_encoder.py.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def make_state_dict(rng, layers=(2, 2, 2, 2), widen=1.0, n_in=3, conv1_t=7, hidden=2048, out_dim=128,
                    projection_head=True, dtype=np.float32, bottleneck=False):
    """weights by the reference's init RULES (models/resnet.py:203-210: Conv3d kaiming_normal_(fan_out, relu),
    BN3d gamma=1 beta=0; Linear / BN1d PyTorch defaults) drawn from a numpy Generator so the GPU box
    regenerates them without torch's RNG.  Keys/shapes = the reference's state_dict."""
    planes = [int(c * widen) for c in (64, 128, 256, 512)]
    sd = {}

    def conv(name, cout, cin, k):
        fan_out = cout * k[0] * k[1] * k[2]
        sd[name] = (rng.standard_normal((cout, cin) + tuple(k)) * math.sqrt(2.0 / fan_out)).astype(dtype)

    def bn(name, c):
        sd[name + ".weight"] = np.ones(c, dtype)
        sd[name + ".bias"] = np.zeros(c, dtype)
        sd[name + ".running_mean"] = np.zeros(c, dtype)
        sd[name + ".running_var"] = np.ones(c, dtype)
        sd[name + ".num_batches_tracked"] = np.zeros((), np.int64)

    def linear(name, cout, cin):
        b = 1.0 / math.sqrt(cin)
        sd[name + ".weight"] = rng.uniform(-b, b, (cout, cin)).astype(dtype)    # kaiming_uniform(a=sqrt(5))
        sd[name + ".bias"] = rng.uniform(-b, b, cout).astype(dtype)

    conv("conv1.weight", planes[0], n_in, (conv1_t, 7, 7))
    bn("bn1", planes[0])
    inp = planes[0]
    for li, (p, nb) in enumerate(zip(planes, layers), start=1):
        for b in range(nb):
            stride = 2 if (li > 1 and b == 0) else 1
            pre = f"layer{li}.{b}"
            if bottleneck:                       # models/resnet.py:58-72: 1x1x1 -> 3x3x3 (stride) -> 1x1x1 to 4 p channels
                conv(pre + ".conv1.weight", p, inp, (1, 1, 1))
                bn(pre + ".bn1", p)
                conv(pre + ".conv2.weight", p, p, (3, 3, 3))
                bn(pre + ".bn2", p)
                conv(pre + ".conv3.weight", 4 * p, p, (1, 1, 1))
                bn(pre + ".bn3", 4 * p)
                if stride != 1 or inp != 4 * p:
                    conv(pre + ".downsample.0.weight", 4 * p, inp, (1, 1, 1))
                    bn(pre + ".downsample.1", 4 * p)
                inp = 4 * p
                continue
            conv(pre + ".conv1.weight", p, inp, (3, 3, 3))
            bn(pre + ".bn1", p)
            conv(pre + ".conv2.weight", p, p, (3, 3, 3))
            bn(pre + ".bn2", p)
            if stride != 1 or inp != p:
                conv(pre + ".downsample.0.weight", p, inp, (1, 1, 1))
                bn(pre + ".downsample.1", p)
            inp = p
    if projection_head:
        linear("fc1", hidden, inp)
        bn("bn_proj", hidden)
        linear("fc2", out_dim, hidden)
    return sd


def to_torch(sd, dtype=torch.float32, requires_grad=False):
    out = {}
    for k, v in sd.items():
        t = torch.as_tensor(np.asarray(v))
        if t.dtype.is_floating_point:
            t = t.to(dtype).clone()
            if requires_grad and "running" not in k:
                t.requires_grad_(True)
        else:
            t = t.clone()
        out[k] = t
    return out


def _bn(x, sd, name, training, momentum=0.1, eps=1e-5):
    y = F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                     sd[name + ".bias"], training, momentum, eps)
    if training:
        sd[name + ".num_batches_tracked"] += 1
    return y


def encoder_forward(sd, x, training, conv1_stride=(1, 2, 2), projection_head=True, taps=None, relu_masks=None, max_pool=False,
                    mask_report=None):
    """x: [B, C, T, H, W].  sd: dict of torch tensors (running stats are updated in place when training).
    taps (optional dict) receives intermediate activations by name.
    relu_masks (optional dict name -> bool tensor, names 'stem', 'layer{L}.{b}.a1', 'layer{L}.{b}', 'head'): the branch every
    ReLU takes is IMPOSED (y = x * mask) instead of decided by the sign of x — a ReLU input within rounding noise of zero goes
    either way in any fp32 run, and a gradient comparison is only well posed between runs that took the same branches.
    mask_report (optional dict, with relu_masks): per ReLU, how the imposed branches differ from the ones THIS run's own
    pre-activation v would take — {numel, flipped = #(mask != (v > 0)), worst = max |v| over the flipped elements, scale = rms(v)} —
    so that a test can bound the imposed masks: a few elements, all within rounding noise of zero (assert_masks_benign)."""
    def relu(v, name):
        if relu_masks is not None and name in relu_masks:
            m = relu_masks[name]
            if mask_report is not None:
                with torch.no_grad():
                    diff = (v > 0) != m
                    n = int(diff.sum())
                    mask_report[name] = dict(numel=v.numel(), flipped=n, worst=float(v[diff].abs().max()) if n else 0.0,
                                             scale=float(v.double().pow(2).mean().sqrt()))
            return v * m.to(v.dtype)
        return F.relu(v)

    kt = sd["conv1.weight"].shape[2]
    x = F.conv3d(x, sd["conv1.weight"], None, conv1_stride, (kt // 2, 3, 3))
    x = relu(_bn(x, sd, "bn1", training), "stem")
    if taps is not None:
        taps["stem"] = x
    if max_pool:                                   # `if not self.no_max_pool: x = self.maxpool(x)` (models/resnet.py:123, 262-263)
        x = F.max_pool3d(x, kernel_size=3, stride=2, padding=1)
    li = 1
    while f"layer{li}.0.conv1.weight" in sd:
        b = 0
        while f"layer{li}.{b}.conv1.weight" in sd:
            pre = f"layer{li}.{b}"
            stride = 2 if (li > 1 and b == 0) else 1
            res = x
            if pre + ".conv3.weight" in sd:          # Bottleneck.forward (models/resnet.py:74-96)
                out = F.conv3d(x, sd[pre + ".conv1.weight"], None, 1, 0)
                out = relu(_bn(out, sd, pre + ".bn1", training), pre + ".a1")
                out = F.conv3d(out, sd[pre + ".conv2.weight"], None, stride, 1)
                out = relu(_bn(out, sd, pre + ".bn2", training), pre + ".a2")
                out = F.conv3d(out, sd[pre + ".conv3.weight"], None, 1, 0)
                out = _bn(out, sd, pre + ".bn3", training)
            else:
                out = F.conv3d(x, sd[pre + ".conv1.weight"], None, stride, 1)
                out = relu(_bn(out, sd, pre + ".bn1", training), pre + ".a1")
                if taps is not None:
                    taps[pre + ".a1"] = out
                out = F.conv3d(out, sd[pre + ".conv2.weight"], None, 1, 1)
                if taps is not None:
                    taps[pre + ".z2"] = out
                out = _bn(out, sd, pre + ".bn2", training)
            if pre + ".downsample.0.weight" in sd:
                res = F.conv3d(x, sd[pre + ".downsample.0.weight"], None, stride, 0)
                res = _bn(res, sd, pre + ".downsample.1", training)
            elif stride != 1 or res.shape[1] != out.shape[1]:
                # shortcut_type 'A' (models/resnet.py:213-222; a state_dict without downsample weights): avg_pool3d(kernel 1,
                # stride) + zero channels, concatenated through `.data` — the branch carries no gradient
                res = F.avg_pool3d(x, kernel_size=1, stride=stride).detach()
                res = torch.cat([res, res.new_zeros((res.shape[0], out.shape[1] - res.shape[1]) + tuple(res.shape[2:]))], 1)
            x = relu(out + res, pre)
            if taps is not None:
                taps[pre] = x
            b += 1
        li += 1
    x = F.adaptive_avg_pool3d(x, 1).flatten(1)
    if taps is not None:
        taps["pooled"] = x
    if not projection_head or "fc1.weight" not in sd:
        return x
    h = F.linear(x, sd["fc1.weight"], sd["fc1.bias"])
    h = relu(_bn(h, sd, "bn_proj", training), "head")
    return F.linear(h, sd["fc2.weight"], sd["fc2.bias"])


def assert_masks_benign(report, max_frac=2e-5, max_rel=1e-4, floor=2):
    """the imposed ReLU branches (encoder_forward(relu_masks=..., mask_report=report)) may differ from the run's own only in a
    handful of elements per layer (at most max(floor, max_frac * numel)) and only where the pre-activation is within rounding
    noise of zero (|v| <= max_rel * rms of the layer's pre-activations).  Returns (total flipped, total elements) for the log."""
    tot = cnt = 0
    for name, r in report.items():
        assert r["flipped"] <= max(floor, max_frac * r["numel"]), (name, r)
        assert r["worst"] <= max_rel * r["scale"], (name, r)
        tot += r["flipped"]
        cnt += r["numel"]
    return tot, cnt


def r3d_to_resnet_keys(sd):
    """state_dict of the reference's R3DNet (models/r3d/r3d.py:126-187) under the 3D-ResNet key names encoder_forward
    walks: conv{L+1}.block1 -> layer{L}.0, conv{L+1}.blocks.i -> layer{L}.{i+1}, *.temporal_spatial_conv.weight -> *.weight,
    downsampleconv / downsamplebn -> downsample.0 / downsample.1.  R3DNet.forward is then
    encoder_forward(mapped, x, training, conv1_stride=(1, 2, 2), projection_head=False) (3x7x7 stem, padding (1, 3, 3))."""
    out = {}
    for k, v in sd.items():
        parts = k.split(".")
        if parts[0] in ("conv2", "conv3", "conv4", "conv5"):
            li = int(parts[0][4:]) - 1
            if parts[1] == "block1":
                pre, rest = f"layer{li}.0", parts[2:]
            else:
                pre, rest = f"layer{li}.{int(parts[2]) + 1}", parts[3:]
            rest = [r for r in rest if r != "temporal_spatial_conv"]
            if rest[0] == "downsampleconv":
                rest = ["downsample", "0"] + rest[1:]
            elif rest[0] == "downsamplebn":
                rest = ["downsample", "1"] + rest[1:]
            out[".".join([pre] + rest)] = v
        else:
            out[".".join(p for p in parts if p != "temporal_spatial_conv")] = v
    return out


def cosine_similarity_rows(x, Y, eps=1e-8):
    """F.cosine_similarity(x[None], Y, dim=1): x.y / (max(|x|, eps) * max(|y|, eps))"""
    xn = x.norm().clamp_min(eps)
    yn = Y.norm(dim=1).clamp_min(eps)
    return (Y @ x) / (xn * yn)


def ntxent_loss(emb, temperature=0.5):
    """loss/triplet_loss.py:97-116"""
    n = emb.shape[0]
    rows = [1 - cosine_similarity_rows(emb[i], emb) for i in range(n)]          # pdist (:429-437)
    sim = 1 - torch.stack(rows, 0)
    sim = sim.masked_fill(torch.eye(n, dtype=torch.bool), 0)                       # diagonal -> 0, not -inf
    sim = sim / temperature
    tgt = (torch.arange(n) + n // 2) % n
    return F.cross_entropy(sim, tgt)


def margin_cosine_loss(anchor, near, far, margin):
    """the LLC term of triplet_train_epoch (online_train.py:317-332): dist_ap = 1 - cos(anc, anc2),
    dist_an = 1 - cos(anc, pos), MarginRankingLoss(margin)(dist_ap, dist_an, target = -1)"""
    d_ap = 1 - F.cosine_similarity(anchor, near, dim=1)
    d_an = 1 - F.cosine_similarity(anchor, far, dim=1)
    return F.margin_ranking_loss(d_ap, d_an, torch.full_like(d_ap, -1.0), margin=margin)


def ntxent_euclid_loss(emb, temperature=0.5):
    """loss/triplet_loss.py:97-116 with dist_metric='euclidean': pdist (:429-437) = F.pairwise_distance(v_i, vectors, eps=0)"""
    n = emb.shape[0]
    rows = [F.pairwise_distance(emb[i], emb, eps=0) for i in range(n)]
    sim = 1 - torch.stack(rows, 0)
    sim = sim.masked_fill(torch.eye(n, dtype=torch.bool), 0)
    sim = sim / temperature
    tgt = (torch.arange(n) + n // 2) % n
    return F.cross_entropy(sim, tgt)


def margin_distance_loss(anchor, near, far, margin, dist_metric='cosine'):
    """the margin terms of triplet_train_epoch (online_train.py:288-360): dist = F.pairwise_distance(a, b, 2) ('euclidean') or
    1 - F.cosine_similarity, MarginRankingLoss(margin)(dist(anchor, near), dist(anchor, far), target = -1)"""
    if dist_metric == 'euclidean':
        d_ap, d_an = F.pairwise_distance(anchor, near, 2), F.pairwise_distance(anchor, far, 2)
    else:
        d_ap, d_an = 1 - F.cosine_similarity(anchor, near, dim=1), 1 - F.cosine_similarity(anchor, far, dim=1)
    return F.margin_ranking_loss(d_ap, d_an, torch.full_like(d_ap, -1.0), margin=margin)


def nce_average(l, ab, y, idx, memory_l, memory_ab, T=0.07, momentum=0.5):
    """NCEAverage.forward with use_softmax=True (loss/NCE_loss.py:26-88).  Banks are updated in place.
    Note the cross-wiring: out_ab uses memory_l, out_l uses memory_ab."""
    B, D = l.shape
    K1 = idx.shape[1]
    w_l = memory_l.index_select(0, idx.reshape(-1)).detach().view(B, K1, D)
    out_ab = torch.bmm(w_l, ab.view(B, D, 1)) / T
    w_ab = memory_ab.index_select(0, idx.reshape(-1)).detach().view(B, K1, D)
    out_l = torch.bmm(w_ab, l.view(B, D, 1)) / T
    with torch.no_grad():
        for mem, f in ((memory_l, l), (memory_ab, ab)):
            pos = mem.index_select(0, y) * momentum + f * (1 - momentum)
            pos = pos / pos.pow(2).sum(1, keepdim=True).pow(0.5)
            mem.index_copy_(0, y, pos)
    return out_l, out_ab


def nce_softmax_loss(x):
    """NCESoftmaxLoss (loss/NCE_loss.py:347-352): CE of [B, K+1] logits against class 0"""
    x = x.squeeze(-1)
    return F.cross_entropy(x, torch.zeros(x.shape[0], dtype=torch.long))


def sgd_step(params, grads, bufs, lr=0.1, momentum=0.5):
    """torch.optim.SGD(lr, momentum) (online_train.py:543): buf = m*buf + g (first step buf = g); p -= lr*buf"""
    for k in params:
        g = grads[k]
        if k not in bufs:
            bufs[k] = g.clone()
        else:
            bufs[k].mul_(momentum).add_(g)
        params[k].data.add_(bufs[k], alpha=-lr)
