"""Batch-sharded inference across the GPUs of one node.

Images are independent through forward, decode and NMS, so the batch is split
contiguously over ranks (one process per GPU) with no data-path collective; the
only exchange is ONE all-gather of the fixed-size, padded per-image detections
(300x6 rows + 300 indices + count = 8.4 KB/image, one flat buffer per rank), which a
caller can overlap with the next batch (``async_op=True``).  ``torch.distributed`` backend
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


MAX_DET = 300   # utils/utils.py:243 (== YFV2_MAX_DET)


def packed_det_buffers(B, device):
    """(dets (B,300,6) f32, idx (B,300) i32, cnt (B,) i32) as views of ONE flat buffer [dets | idx | cnt] of 4-byte words:
    what `Engine.new_det_buffers` hands out, so that a rank's whole result travels in one collective."""
    B = int(B)
    flat = torch.empty(B * (MAX_DET * 6 + MAX_DET + 1), dtype=torch.float32, device=device)
    dets = flat[:B * MAX_DET * 6].view(B, MAX_DET, 6)
    idx = flat[B * MAX_DET * 6:B * MAX_DET * 7].view(torch.int32).view(B, MAX_DET)
    cnt = flat[B * MAX_DET * 7:].view(torch.int32)
    dets._yfv2_packed = flat
    return dets, idx, cnt


def _packed_of(dets, idx, cnt):
    """the flat buffer behind a (dets, idx, cnt) triple made by packed_det_buffers, or None"""
    flat = getattr(dets, "_yfv2_packed", None)
    if flat is None or dets.dim() != 3:
        return None
    B = dets.shape[0]
    ok = (flat.numel() == B * (MAX_DET * 7 + 1) and dets.data_ptr() == flat.data_ptr() and
          idx.data_ptr() == flat.data_ptr() + 4 * B * MAX_DET * 6 and cnt.data_ptr() == flat.data_ptr() + 4 * B * MAX_DET * 7 and
          tuple(idx.shape) == (B, MAX_DET) and tuple(cnt.shape) == (B,))
    return flat if ok else None


def _unpack_gathered(g, W, B):
    """[rank][dets | idx | cnt] -> (W*B,300,6), (W*B,300), (W*B): three small device copies"""
    g = g.view(W, -1)
    dets = g[:, :B * MAX_DET * 6].reshape(W * B, MAX_DET, 6)
    idx = g[:, B * MAX_DET * 6:B * MAX_DET * 7].contiguous().view(torch.int32).view(W * B, MAX_DET)
    cnt = g[:, B * MAX_DET * 7:].contiguous().view(torch.int32).view(W * B)
    return dets, idx, cnt


class GatherWork:
    """Handle of an asynchronous gather_detections: wait() orders the current stream (NCCL / RCCL) or the host (gloo)
    behind the collective and returns the gathered (dets, idx, cnt)."""

    def __init__(self, works, finish):
        self._works, self._finish, self._result = works, finish, None

    def wait_host(self):
        """Block the HOST until the collective has completed, without enqueueing anything on the caller's stream (a stream
        wait costs a ~10 us bubble on the compute stream; a collective issued two steps ago finished long before)."""
        import time
        for w in self._works:
            spins = 0
            while not w.is_completed():      # normally true on the first look: the collective was issued several steps ago
                spins += 1
                if spins > 64:               # still running: give the core away between looks instead of burning it
                    time.sleep(20e-6)
        self._works = []

    def wait(self, unpack=True):
        """unpack=False only orders behind the collective (the packed receive buffer passed as `out` then holds
        [rank][dets | idx | cnt]; `rank_views` reads it without copies) - the per-step form of bench.py"""
        for w in self._works:
            w.wait()
        self._works = []
        if not unpack:
            return None
        if self._result is None:
            self._result = self._finish()
        return self._result


def rank_views(g, W, B):
    """zero-copy views into a packed receive buffer: [(dets, idx, cnt) of rank 0, of rank 1, ...]"""
    g = g.view(W, -1)
    return [(g[r, :B * MAX_DET * 6].view(B, MAX_DET, 6), g[r, B * MAX_DET * 6:B * MAX_DET * 7].view(torch.int32).view(B, MAX_DET),
             g[r, B * MAX_DET * 7:].view(torch.int32)) for r in range(W)]


def gather_detections(dets, idx, cnt, group=None, force=False, async_op=False, out=None):
    """All-gather equally-sized per-rank results -> (W*B,300,6), (W*B,300), (W*B) on every rank.
    Buffers from `Engine.new_det_buffers` / `packed_det_buffers` travel as ONE collective (8.4 KB per image); any other
    triple as three.  With one rank the collective is skipped unless ``force`` (used to exercise RCCL at N=1).
    ``async_op=True`` returns a GatherWork instead: the collective runs on the backend's own stream while the caller
    enqueues the next batch; ``out`` (packed path) is a reusable flat receive buffer of W * B * 2101 float32 words."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        res = (dets, idx, cnt)
        return GatherWork([], lambda: res) if async_op else res
    W = dist.get_world_size(group)
    flat = _packed_of(dets, idx, cnt)
    if flat is not None:
        B = dets.shape[0]
        g = out if out is not None else torch.empty(W * flat.numel(), dtype=flat.dtype, device=flat.device)
        if g.numel() != W * flat.numel() or g.dtype != flat.dtype or g.device != flat.device:
            raise ValueError("gather_detections: `out` must be a flat float32 tensor of %d words on %s" % (W * flat.numel(), flat.device))
        work = dist.all_gather_into_tensor(g, flat, group=group, async_op=async_op)
        if async_op:
            return GatherWork([work], lambda: _unpack_gathered(g, W, B))
        return _unpack_gathered(g, W, B)
    outs, works = [], []
    for t in (dets, idx, cnt):
        t = t.contiguous()
        g = torch.empty((W * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        works.append(dist.all_gather_into_tensor(g, t, group=group, async_op=async_op))
        outs.append(g)
    if async_op:
        return GatherWork(works, lambda: tuple(outs))
    return tuple(outs)


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


def average_gradients_(flat, group=None, force=False):
    """Data-parallel TRAINING across the GPUs of one node (SURVEY.md 8(e) "Training"): every rank runs train.py:101-110 on
    its shard of the batch (batch-statistics BatchNorm per rank, as torch's DistributedDataParallel without SyncBatchNorm
    does) and the gradients are averaged by ONE all-reduce over the flat bucket the backward kernels wrote them into
    (``Detector`` keeps all 225 gradients in one 243 095-float buffer, 0.97 MB: a single latency-bound collective over xGMI,
    no bucketing, no copies).  In place; returns ``flat``.  No-op without an initialised process group or on one rank
    (``force=True`` still issues the collective: how the RCCL path is exercised on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return flat
    W = dist.get_world_size(group)
    if W == 1 and not force:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if W > 1:
        flat.mul_(1.0 / W)
    return flat
