"""
The hot loops of the reference's online_train.py, with the same function names, argument order and step
structure, on the HIP path:

    diff(x)                                              <- online_train.py:228-230
    triplet_train_epoch(train_loader, model, criterion, optimizer, epoch, cfg, cuda, device, is_master_proc)
                                                         <- online_train.py:245-434  (default and LOCAL_LOCAL_CONTRAST branches)
    contrastive_train_epoch(train_loader, model, criterion_1, criterion_2, contrast, optimizer, epoch, cfg, cuda,
                            device, is_master_proc)      <- online_train.py:136-225
    iterative_cluster_step(args, cfg, encoder, eval_train_loader, epoch, cuda, device, is_master_proc)
                                                         <- online_train.py:605-662 (embed -> fit_cluster -> vid_clusters.txt -> barrier)

What stays the reference's: one forward over cat(views), slicing, `loss = criterion + lambda * margin term`,
zero_grad / backward / step, the two scalar all-reduces, the text logs.  What changes: the encoder, the
NT-Xent / memory-bank losses, the margin-ranking term on cosine distances, and the clustering run on the GPU;
the per-step `.item()` host syncs of the reference (`:389,392`) are batched to one per LOG_INTERVAL.
`cfg` is any object with the reference's key names (cfg.LOSS.LOCAL_LOCAL_CONTRAST, cfg.DATASET.SAMPLING_STRATEGY,
cfg.NUM_GPUS, cfg.TRAIN.LOG_INTERVAL, cfg.OUTPUT_PATH, cfg.ITERCLUSTER.*): fvcore's CfgNode or a SimpleNamespace tree.
Branches the shipped SLIC configs never take (RELATIVE_SPEED_PERCEPTION, INTRA_NEGATIVE, slowfast inputs,
`intra_neg` / `moco`) raise — the reference's own code for them references undefined names.
"""
import os
import time

import torch

from .loss.triplet_loss import margin_cosine_loss, margin_distance_loss   # the LLC / RSP / intra-negative terms share the margin kernels
from .misc import distributed_helper as du_helper

modality = 'res'        # module-level switch of the reference (online_train.py:36)


class AverageMeter(object):
    """models/model_utils.py:214-229"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def diff(x):
    """residual-frame view (online_train.py:228-230); pure data movement, stays a torch op"""
    shift_x = torch.roll(x, 1, 2)
    return ((x - shift_x) + 1) / 2


def _flag(node, name, default=False):
    return bool(getattr(node, name, default))


def triplet_train_epoch(train_loader, model, criterion, optimizer, epoch, cfg, cuda, device, is_master_proc=True):
    losses = AverageMeter()
    running_n_triplets = AverageMeter()
    world_size = du_helper.get_world_size()
    model.train()
    start = time.time()
    pending = []        # (loss tensor, batch_size_world tensor) kept on the device until the next log line
    for batch_idx, (inputs, targets, idx) in enumerate(train_loader):
        # a third clip per sample and a margin term on it (online_train.py:256-360), in the reference's order of precedence:
        #   RELATIVE_SPEED_PERCEPTION: fast positive, d(anc, pos) + 0.1 < d(anc, fast_pos), weight 1
        #   LOCAL_LOCAL_CONTRAST     : second anchor clip, d(anc, anc2) + LOCAL_LOCAL_MARGIN < d(anc, pos), weight LOCAL_LOCAL_WEIGHT
        #   INTRA_NEGATIVE           : intra-video negative, d(anc, intra_neg) + 0.04 < d(anc, pos), weight 0.4 (as the reference has it)
        third = ("rsp" if _flag(cfg.LOSS, "RELATIVE_SPEED_PERCEPTION") else "llc" if _flag(cfg.LOSS, "LOCAL_LOCAL_CONTRAST") else
                 "intra" if _flag(cfg.LOSS, "INTRA_NEGATIVE") else None)
        if third:
            anchor, positive, extra = inputs
            extra = extra.to(device)
        else:
            anchor, positive = inputs
        anchor, positive = anchor.to(device), positive.to(device)
        a_target, p_target = targets
        batch_size = torch.tensor(anchor.size(0)).to(device)
        targets = torch.cat((a_target, p_target), 0).to(device)
        b = anchor.size(0)
        if third:
            outputs = model(torch.cat((anchor, positive, extra), 0))        # ONE forward: BN stats over all 3b clips
            out_anchor_positive = outputs[:b * 2]
            out_anc, out_pos, out_extra = outputs[:b], outputs[b:2 * b], outputs[2 * b:3 * b]
            triplet_loss, n_triplets = criterion(out_anchor_positive, targets, sampling_strategy=cfg.DATASET.SAMPLING_STRATEGY)
            metric = cfg.LOSS.DIST_METRIC
            # MarginRankingLoss(margin)(dist_ap, dist_an, -1) = mean max(0, dist_ap - dist_an + margin)
            if third == "rsp":
                loss = triplet_loss + margin_distance_loss(out_anc, out_pos, out_extra, 0.1, metric) * 1.0
            elif third == "llc":
                loss = triplet_loss + margin_distance_loss(out_anc, out_extra, out_pos, cfg.LOSS.LOCAL_LOCAL_MARGIN, metric) * \
                    cfg.LOSS.LOCAL_LOCAL_WEIGHT
            else:
                loss = triplet_loss + margin_distance_loss(out_anc, out_extra, out_pos, 0.04, metric) * 0.4
        else:
            outputs = model(torch.cat((anchor, positive), 0))
            loss, n_triplets = criterion(outputs, targets, sampling_strategy=cfg.DATASET.SAMPLING_STRATEGY)
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        loss = loss.detach()
        if cfg.NUM_GPUS > 1:
            [loss] = du_helper.all_reduce([loss], avg=True)
            [batch_size_world] = du_helper.all_reduce([batch_size], avg=False)
        else:
            batch_size_world = batch_size
        pending.append((loss, batch_size_world))
        running_n_triplets.update(n_triplets)
        log_now = ((batch_idx + 1) * world_size) % cfg.TRAIN.LOG_INTERVAL == 0
        if log_now or batch_idx + 1 == len(train_loader):
            for l, bs in pending:                                           # one host sync per log interval
                losses.update(l.item(), bs.item())
            pending = []
            if is_master_proc and log_now:
                print('Train Epoch: {} [{}/{} | {:.1f}%]\tLoss: {:.4f} ({:.4f}) \tN_Triplets: {:.1f}'.format(
                    epoch, losses.count, len(train_loader.dataset), 100. * (losses.count / len(train_loader.dataset)),
                    losses.val, losses.avg, running_n_triplets.avg))
    for l, bs in pending:
        losses.update(l.item(), bs.item())
    if is_master_proc:
        print('\nTrain set: Average loss: {:.4f}\n'.format(losses.avg))
        print('epoch:{} runtime:{}'.format(epoch, (time.time() - start) / 3600))
        _append_log(cfg, 'train_loss_and_acc.txt', 'epoch:{} runtime:{} {:.4f}\n'.format(
            epoch, round((time.time() - start) / 3600, 2), losses.avg))
    return losses.avg


def contrastive_train_epoch(train_loader, model, criterion_1, criterion_2, contrast, optimizer, epoch, cfg, cuda, device,
                            is_master_proc=True):
    losses = AverageMeter()
    world_size = du_helper.get_world_size()
    model.train()
    contrast.train()
    start = time.time()
    pending = []
    from .loss.NCE_loss import NCEAverage, NCESoftmaxLoss
    fused_step = (isinstance(contrast, NCEAverage) and contrast.use_softmax and type(criterion_1) is NCESoftmaxLoss and
                  type(criterion_2) is NCESoftmaxLoss)
    for batch_idx, (inputs, labels, index) in enumerate(train_loader):
        view1 = inputs[0]
        view2 = inputs[1] if modality == 'rgb' else diff(view1)
        batch_size = torch.tensor(view1.size(0)).to(device)
        view1, view2 = view1.to(device), view2.to(device)
        index = index.to(device)
        feat_1 = model(view1)                # two separate forwards: BN statistics per view (online_train.py:175-176)
        feat_2 = model(view2)
        if fused_step:
            # the same sum, banks updated the same way, as three launches (loss/NCE_loss.py: NCEAverage.softmax_loss)
            loss, out_1, out_2 = contrast.softmax_loss(feat_1, feat_2, index)
        else:
            out_1, out_2 = contrast(feat_1, feat_2, index)
            view1_loss = criterion_1(out_1)
            view2_loss = criterion_2(out_2)
            loss = view1_loss + view2_loss
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        loss = loss.detach()
        if cfg.NUM_GPUS > 1:
            [loss] = du_helper.all_reduce([loss], avg=True)
            [batch_size_world] = du_helper.all_reduce([batch_size], avg=False)
        else:
            batch_size_world = batch_size
        pending.append((loss, batch_size_world))
        log_now = ((batch_idx + 1) * world_size) % cfg.TRAIN.LOG_INTERVAL == 0
        if log_now or batch_idx + 1 == len(train_loader):
            for l, bs in pending:
                losses.update(l.item(), bs.item())
            pending = []
            if is_master_proc and log_now:
                print('Train Epoch: {} [{}/{} | {:.1f}%]\tLoss: {:.4f} ({:.4f})'.format(
                    epoch, losses.count, len(train_loader.dataset), 100. * (losses.count / len(train_loader.dataset)),
                    losses.val, losses.avg))
    if is_master_proc:
        print('\nTrain set: Average loss: {:.4f}\n'.format(losses.avg))
        _append_log(cfg, 'train_loss_and_acc.txt', 'epoch:{} runtime:{} {:.4f}\n'.format(
            epoch, round((time.time() - start) / 3600, 2), losses.avg))
    return losses.avg


def _append_log(cfg, name, line):
    out = getattr(cfg, "OUTPUT_PATH", None)
    if not out:
        return
    d = os.path.join(out, 'tnet_checkpoints')
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, name), "a") as f:
        f.write(line)


def _dataset_order(n, idxs, cluster_labels):
    """cluster assignments in the unshuffled order of the dataset (online_train.py:649-651): slot idxs[i] <- label i, later
    duplicates (DistributedSampler padding) overwrite earlier ones; slots the loader never produced (drop_last) stay -1"""
    import numpy as np
    order = np.full(n, -1, dtype=np.int32)
    idxs = np.asarray(idxs, dtype=np.int64).reshape(-1)
    lab = np.asarray(cluster_labels).reshape(-1).astype(np.int32)
    if len(idxs) != len(lab):
        raise ValueError("one dataset index per clustered row: {} vs {}".format(len(idxs), len(lab)))
    if len(idxs) and (idxs.min() < 0 or idxs.max() >= n):
        raise ValueError("dataset index outside [0, {})".format(n))
    order[idxs] = lab               # numpy assigns in order: the last duplicate wins, like the reference's loop
    return order


def _cluster_sharded(cfg):
    """the sharded route of SURVEY.md §8e (configs[2]): every rank keeps its rows, k-means runs over the process group"""
    return (cfg.NUM_GPUS > 1 and torch.distributed.is_available() and torch.distributed.is_initialized()
            and cfg.ITERCLUSTER.METHOD in ('kmeans', 'spherical_kmeans') and bool(getattr(cfg.ITERCLUSTER, "SHARDED", True)))


def iterative_cluster_step(args, cfg, encoder, eval_train_loader, epoch, cuda=True, device=None, is_master_proc=True,
                           kmeans_kernels=None):
    """online_train.py:605-662: embeddings of the whole train set -> fit_cluster -> NMI/AMI logs -> vid_clusters.txt in
    the dataset's unshuffled order -> barrier.

    Returns (labels, NMI or None): labels = np.int32 [len(dataset)] in DATASET order (the content of vid_clusters.txt,
    -1 where the eval loader produced no row, e.g. drop_last), identical on every rank.

    NUM_GPUS > 1 with k-means (cfg.ITERCLUSTER.SHARDED, default on): the extraction keeps each rank's [N/W, D] rows on
    its GPU (no per-batch all_gather + D2H, evaluate.py:189-193), fit_cluster runs row-sharded over the process group
    (ONE fp64 all-reduce of [K*D sums | K counts | n_changed] per Lloyd iteration over RCCL), and the only other exchange is one int32 all-gather of
    (label, dataset index, true label).  FINCH — and SHARDED = False — keep the reference's shape: gather to every rank,
    cluster on rank 0, and the dataset-ordered labels are broadcast (which is also the barrier of :662).
    `kmeans_kernels`: another kernel provider for fit_cluster (the tests of the multi-process control flow pass a CPU one
    as an argument; the product passes nothing and runs the HIP kernels)."""
    import numpy as np
    from .clustering.cluster_masks import fit_cluster
    from .evaluate import get_embeddings_and_labels
    n_data = len(eval_train_loader.dataset)
    sharded = _cluster_sharded(cfg)
    if is_master_proc:
        print('\n=> Computing embeddings')
    start_time = time.time()
    embeddings, true_labels, idxs = get_embeddings_and_labels(args, cfg, encoder, cuda, device, eval_train_loader,
                                                              split='train', is_master_proc=is_master_proc,
                                                              gather=not sharded)
    if is_master_proc:
        print('Time to get embeddings: {:.2f}s'.format(time.time() - start_time))
    order, NMI = None, None
    err = None
    if sharded:
        if is_master_proc:
            print('\n=> Clustering')
            print('embeddings shape (this rank)', tuple(embeddings.shape))
        start_time = time.time()
        pg = torch.distributed.group.WORLD
        # A rank whose fit_cluster raises (bad K for its shard, out of memory) must not leave the others waiting in the label
        # all-gather below: every rank reports, and all raise together.  (A rank whose local half of a Lloyd iteration raises keeps
        # entering the loop's collectives with a poisoned payload, and every rank raises at that iteration: kmeans_hip.KMeans;
        # only a peer that DIES strands the others until the process group's / SLIC_COMM_TIMEOUT_MS timeout.)
        def all_or_raise(failure, what):
            flag = torch.tensor([0 if failure is None else 1], dtype=torch.int32, device=embeddings.device)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX, group=pg)
            if int(flag.item()):
                raise RuntimeError(f"sharded fit_cluster: {what} failed on a rank (this rank: {failure!r})") from failure

        failure = None
        try:                                     # what can fail on ONE rank before the first collective of the fit
            if kmeans_kernels is not None:
                kmeans_kernels.check()
            else:
                from .clustering.kmeans_hip import HipKernels
                HipKernels().check()
            assert embeddings.dim() == 2 and embeddings.shape[0] > 0, "this rank holds no embeddings"
        except Exception as e:
            failure = e
        all_or_raise(failure, "the device / shard check")
        local, failure = None, None
        try:
            local = fit_cluster(embeddings, cfg.ITERCLUSTER.METHOD, cfg.ITERCLUSTER.K, cfg.ITERCLUSTER.L2_NORMALIZE,
                                getattr(cfg.ITERCLUSTER, "FINCH_PARTITION", 0), process_group=pg, kernels=kmeans_kernels)
        except Exception as e:
            failure = e
        all_or_raise(failure, "the clustering")
        trip = torch.from_numpy(np.stack([np.asarray(local, np.int32), np.asarray(idxs, np.int32),
                                          np.asarray(true_labels, np.int32)])).to(embeddings.device)
        cluster_labels, idxs, true_labels = (a.reshape(-1) for a in _all_gather_rows(trip, pg))
        if is_master_proc:
            print('Time to cluster: {:.2f}s'.format(time.time() - start_time))
        order = _dataset_order(n_data, idxs, cluster_labels)       # every rank holds all triples: no broadcast needed
    elif is_master_proc:
        print('\n=> Clustering')
        start_time = time.time()
        print('embeddings shape', embeddings.size())
        try:
            cluster_labels = fit_cluster(embeddings, cfg.ITERCLUSTER.METHOD, cfg.ITERCLUSTER.K, cfg.ITERCLUSTER.L2_NORMALIZE,
                                         getattr(cfg.ITERCLUSTER, "FINCH_PARTITION", 0),
                                         **({"kernels": kmeans_kernels} if kmeans_kernels is not None else {}))
            print('Time to cluster: {:.2f}s'.format(time.time() - start_time))
            order = _dataset_order(n_data, idxs, cluster_labels)
        except Exception as e:              # the other ranks are about to enter a collective: tell them instead of hanging them
            if cfg.NUM_GPUS <= 1:
                raise
            err = e
    if is_master_proc and order is not None:
        try:                                     # O(N) contingency-table metrics stay on the host
            from sklearn.metrics import adjusted_mutual_info_score, normalized_mutual_info_score
            NMI = normalized_mutual_info_score(true_labels, cluster_labels)
            AMI = adjusted_mutual_info_score(true_labels, cluster_labels)
            print('NMI between true labels and cluster assignments: {:.3f}'.format(NMI))
            print('AMI between true labels and cluster assignments: {:.3f}\n'.format(AMI))
            _append_log(cfg, 'NMIs.txt', 'epoch:{} {:.3f}\n'.format(epoch, NMI))
            _append_log(cfg, 'AMIs.txt', 'epoch:{} {:.3f}\n'.format(epoch, AMI))
            if getattr(cfg.ITERCLUSTER, "ADAPTIVEP", False):
                cfg.DATASET.POSITIVE_SAMPLING_P = float(1.0 - NMI)
        except ImportError:
            pass
        # one label per line, unshuffled dataset order (online_train.py:654-657); a slot the loader never produced is
        # written as the reference writes it ('None')
        cluster_output_path = os.path.join(cfg.OUTPUT_PATH, 'vid_clusters.txt')
        with open(cluster_output_path, "w") as f:
            for label in order:
                f.write('{}\n'.format(label if label >= 0 else None))
        print('Saved cluster labels to', cluster_output_path)
    if cfg.NUM_GPUS > 1 and not sharded:
        # besides the text file (kept for the reference's dataset code, which re-parses it on every
        # rank), hand the labels to the other ranks as one int32 broadcast, so a caller can rebuild its sampler
        # without touching the filesystem; the broadcast is also the barrier of online_train.py:662
        send = order if err is None else np.full(n_data, _FAILED, np.int32)
        order = broadcast_cluster_labels(send, n_data, device, is_master_proc)
        if err is not None:
            raise err
        if len(order) and order[0] == _FAILED:
            raise RuntimeError("clustering failed on the master process")
    elif sharded:
        torch.distributed.barrier()          # the file is complete before any rank re-reads it (online_train.py:662)
    return order, NMI


_FAILED = -2          # first slot of the broadcast when the master could not produce labels


def _all_gather_rows(t, pg):
    """[r, n_local] int32 on every rank (n_local may differ) -> numpy [r, sum n_local], rank-major along the columns"""
    import numpy as np
    W = torch.distributed.get_world_size(pg)
    n = torch.tensor([t.shape[1]], dtype=torch.int64, device=t.device)
    sizes = torch.empty(W, dtype=torch.int64, device=t.device)
    torch.distributed.all_gather_into_tensor(sizes, n, group=pg)
    sizes = sizes.cpu().tolist()
    mx = max(sizes)
    pad = torch.zeros(t.shape[0], mx, dtype=t.dtype, device=t.device)
    pad[:, : t.shape[1]] = t
    out = torch.empty(W * pad.numel(), dtype=t.dtype, device=t.device)
    torch.distributed.all_gather_into_tensor(out, pad.reshape(-1), group=pg)
    out = out.view(W, t.shape[0], mx).cpu().numpy()
    return np.concatenate([out[r, :, : sizes[r]] for r in range(W)], axis=1)


def broadcast_cluster_labels(cluster_labels, n, device, is_master_proc, src=0):
    """dataset-ordered labels (np.ndarray[int] of length n on the master, anything elsewhere) -> the same np.ndarray[int32]
    on every rank.  The master validates BEFORE the collective and still takes part in it when its input is unusable
    (first slot = -2), so a bad array cannot strand the other ranks inside dist.broadcast."""
    import numpy as np
    dev = device if device is not None else ("cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
    buf = torch.empty(n, dtype=torch.int32, device=dev)
    bad = None
    if is_master_proc:
        try:
            lab = np.asarray(cluster_labels, dtype=np.int32).reshape(-1)
            if lab.shape != (n,):
                raise ValueError("one label per dataset item: got {} for a dataset of {}".format(lab.shape, n))
        except (TypeError, ValueError) as e:
            bad = e
            lab = np.full(n, _FAILED, np.int32)
        buf.copy_(torch.from_numpy(lab))
    torch.distributed.broadcast(buf, src=src)
    if bad is not None:
        raise bad
    return buf.cpu().numpy()
