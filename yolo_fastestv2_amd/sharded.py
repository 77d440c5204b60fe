"""Batch-sharded inference across the GPUs of one node.

Images are independent through forward, decode and NMS, so the batch is split
contiguously over ranks (one process per GPU) with no data-path collective; the
only exchange is one all-gather of the fixed-size, padded per-image detections
(count + 300x6 rows + 300 indices = 8.4 KB/image).  ``torch.distributed`` backend
"nccl" is RCCL on ROCm (xGMI); the same code runs under "gloo" on CPU tensors,
which is how the N>1 path is tested without GPUs.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world_size):
    """Contiguous split; the first ``n_items % world_size`` ranks get one extra."""
    base, rem = divmod(int(n_items), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_detections(dets, idx, cnt, group=None, force=False):
    """All-gather equally-sized per-rank results -> (W*B,300,6), (W*B,300), (W*B) on every rank.
    With one rank the collective is skipped unless ``force`` (used to exercise RCCL at N=1)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return dets, idx, cnt
    W = dist.get_world_size(group)
    out = []
    for t in (dets, idx, cnt):
        t = t.contiguous()
        g = torch.empty((W * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(g, t, group=group)
        out.append(g)
    return tuple(out)


def gather_decoded(decoded, group=None, force=False):
    """Debug / parity mode (BASELINE.json's wording "all-gather of decoded boxes"): all-gather the
    per-rank decoded tensor (B_local, rows, 5+classes) -> (W*B_local, rows, 5+classes) on every rank.  158 MB per rank at
    256 images - about a millisecond per xGMI link, comparable to the whole forward - which is why the product path
    (``detect_sharded``) gathers the 8.4 KB/image padded detections instead."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return decoded
    W = dist.get_world_size(group)
    decoded = decoded.contiguous()
    g = torch.empty((W * decoded.shape[0],) + tuple(decoded.shape[1:]), dtype=decoded.dtype, device=decoded.device)
    dist.all_gather_into_tensor(g, decoded, group=group)
    return g


def detect_sharded(engine, x_local, conf_thres, iou_thres, group=None, out=None):
    """This rank's shard through forward+decode+NMS, then the all-gather.
    Every rank must pass the same local batch size (pad the last shard)."""
    dets, idx, cnt = engine.detect(x_local, conf_thres, iou_thres, out=out)
    return gather_detections(dets, idx, cnt, group)
