"""Drop-ins for the hot-path functions of the reference's ``utils/utils.py``:
``handel_preds`` (:303-358) and ``non_max_suppression`` (:232-296), same
signatures and return types, executed by libyfv2's HIP kernels.  Also
``load_datafile`` (:13-65), the ``.data`` config reader those callers need, and the
evaluation loop around the path: ``get_batch_statistics`` (:194-230,
one kernel launch per batch), ``evaluation`` (:360-397, device-resident until the last line)
and the dataset-level host arithmetic ``ap_per_class`` / ``compute_ap`` (:110-192, numpy like
the reference: a few thousand float64 operations once per evaluation).
"""
import os

import torch

from ..engine import get_engine, unpack_detections

_LIST_KEYS = ("anchors", "steps")
_STR_KEYS = ("model_name", "val", "train", "names", "pre_weights")
_INT_KEYS = ("epochs", "batch_size", "classes", "width", "height", "anchor_num", "subdivisions")
_FLOAT_KEYS = ("learning_rate",)


def load_datafile(data_path):
    """Parse a darknet-style ``.data`` file into the cfg dict the reference uses:
    ``[section]`` lines and blank lines are skipped, ``key=value`` typed by key."""
    assert os.path.exists(data_path), "config file %s not found" % data_path
    cfg = {k: None for k in _LIST_KEYS + _STR_KEYS + _INT_KEYS + _FLOAT_KEYS}
    with open(data_path, "r") as f:
        for line in f:
            if line == "\n" or line[0] == "[":
                continue
            key, _, val = line.strip().partition("=")
            if key in _INT_KEYS:
                cfg[key] = int(val)
            elif key in _STR_KEYS:
                cfg[key] = val
            elif key in _FLOAT_KEYS:
                cfg[key] = float(val)
            elif key in _LIST_KEYS:
                cfg[key] = [float(v) for v in val.split(",")]
            else:
                print("%s: unknown key %r ignored" % (data_path, key))
    return cfg


def handel_preds(preds, cfg, device):
    """6-tuple of NCHW logits (on the GPU) -> (B, 1815, 5+classes) fp32 **CPU** tensor,
    exactly the reference's contract (it allocates with torch.zeros(...) on the CPU,
    utils.py:328).  The GPU copy is kept on the returned tensor (``_yfv2_dev``) so
    that ``non_max_suppression`` can skip the upload when handed the same object."""
    preds = list(preds)
    if len(preds) != 6:
        raise ValueError("expected the 6-tuple returned by Detector.forward, got %d tensors" % len(preds))
    p0 = preds[0]
    if p0.device.type != "cuda":
        raise RuntimeError("handel_preds: logits must live on the MI355X (no CPU path)")
    eng = getattr(p0, "_yfv2_engine", None)
    if eng is None or eng.device != p0.device:
        eng = get_engine(p0.device, cfg["height"], cfg["width"], preds[2].shape[1], cfg["anchor_num"])
    eng.set_anchors(cfg["anchors"])
    dev = eng.decode([p.detach().float() for p in preds])
    out = dev.cpu()
    eng.check_finite("handel_preds (the forward that produced these logits)")   # the host has just waited for the device: free
    out._yfv2_dev = (dev, eng, out._version, _host_checksum(out))
    return out


_CACHE_MAX_BYTES = 8 << 20


def _host_checksum(t):
    """What non_max_suppression compares before it trusts the device copy handel_preds left on its CPU result: the reference's callers
    may edit that tensor in place - through torch (bumps `_version`) or through `t.numpy()` (does not).  Two wrap-around integer sums
    over the raw bits (every 32-bit word; every 64-bit pair of words): any edit of a single element changes both, and an edit that
    keeps both is not something a caller produces by accident.  Tensors beyond 8 MB (more than 12 images of 1815 x 85) are not
    cached at all: summing 158 MB on the host costs as much as uploading them."""
    if t.numel() * t.element_size() > _CACHE_MAX_BYTES or not t.is_contiguous() or t.dtype != torch.float32:
        return None
    w = t.view(-1).view(torch.int32)
    s32 = int(w.sum(dtype=torch.int64))
    s64 = int(w[: w.numel() & ~1].view(torch.int64).sum()) if w.numel() >= 2 else 0
    return (t.data_ptr(), t.numel(), s32, s64)


def non_max_suppression(prediction, conf_thres=0.3, iou_thres=0.45, classes=None):
    """(B, rows, 5+classes) -> list of B CPU fp32 tensors (n_i, 6): x1,y1,x2,y2,conf,cls in
    descending conf; empty images give (0, 6).  No 1 s wall-clock abort (utils.py:245,292-294)."""
    rows, _ = nms_with_indices(prediction, conf_thres, iou_thres, classes)
    return rows


def nms_with_indices(prediction, conf_thres=0.3, iou_thres=0.45, classes=None):
    """non_max_suppression that also returns, per image, the index of every
    survivor in the decode row order (SURVEY.md 8(b) 'survivor indices')."""
    cached = getattr(prediction, "_yfv2_dev", None)
    if cached is not None and cached[2] == prediction._version and cached[3] is not None and cached[3] == _host_checksum(prediction):
        dev, eng = cached[0], cached[1]           # the very tensor handel_preds returned, unedited (in-place torch ops AND numpy-side writes are seen)
    else:
        if not torch.cuda.is_available():
            raise RuntimeError("non_max_suppression: no MI355X visible (there is no CPU path)")
        device = prediction.device if prediction.device.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
        dev = prediction.detach().to(device, torch.float32)
        # rows = 3*(H/16*W/16 + H/32*W/32); the NMS kernel only needs (rows, classes)
        eng = _engine_for_rows(device, dev.shape[1], dev.shape[2] - 5)
    dets, idx, cnt = eng.nms(dev, conf_thres, iou_thres, classes)
    return unpack_detections(dets, idx, cnt)


def get_batch_statistics(outputs, targets, iou_threshold, device=None):
    """utils/utils.py:194-230 with the same signature and return structure: a list with one
    [true_positives (np.float64 array), pred_scores (tensor), pred_labels (tensor)] entry per non-None output.
    The per-detection matching loop (the part that dominates evaluation() once the model is fast) runs as one
    kernel launch for the whole batch; outputs are the (n_i, 6) tensors non_max_suppression returned."""
    import numpy as np
    if not torch.cuda.is_available():
        raise RuntimeError("get_batch_statistics: no MI355X visible (there is no CPU path)")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None or torch.device(device).type != "cuda" else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    B = len(outputs)
    eng = get_engine(dev, 32, 32, 1, 3)      # the statistics kernel uses no workspace: the smallest handle there is (a few KB)
    MAXD = 300
    dets = torch.zeros((B, MAXD, 6), dtype=torch.float32)
    cnt = torch.zeros((B,), dtype=torch.int32)
    for i, o in enumerate(outputs):
        if o is None:
            continue
        n = min(int(o.shape[0]), MAXD)
        dets[i, :n] = o[:n].detach().to("cpu", torch.float32)
        cnt[i] = n
    tp = eng.batch_statistics(dets.to(dev), cnt.to(dev), torch.as_tensor(targets), iou_threshold).cpu().numpy()
    metrics = []
    for i, o in enumerate(outputs):
        if o is None:
            continue
        n = int(o.shape[0])
        t = np.zeros(n)
        t[:min(n, MAXD)] = tp[i, :min(n, MAXD)]
        metrics.append([t, o[:, 4], o[:, -1]])
    return metrics


def compute_ap(recall, precision):
    """utils/utils.py:110-134: area under the precision envelope, summed where recall changes."""
    import numpy as np
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([0.0], precision, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]      # running max from the right = the reference's backward loop
    k = np.flatnonzero(mrec[1:] != mrec[:-1])
    return np.sum((mrec[k + 1] - mrec[k]) * mpre[k + 1])


def ap_per_class(tp, conf, pred_cls, target_cls):
    """utils/utils.py:136-192, same arguments and the same 4-tuple (mean precision, mean recall, mean AP, mean F1
    over the classes that occur in ``target_cls``).  Detections are ranked with the same ``np.argsort(-conf)`` call,
    so equal-confidence ties fall exactly as they do in the reference."""
    import numpy as np
    tp, conf, pred_cls, target_cls = np.asarray(tp), np.asarray(conf), np.asarray(pred_cls), np.asarray(target_cls)
    rank = np.argsort(-conf)
    tp, pred_cls = tp[rank], pred_cls[rank]
    p, r, ap = [], [], []
    for c in np.unique(target_cls):
        mine = pred_cls == c
        n_gt = (target_cls == c).sum()
        if not mine.any():          # a ground-truth class nobody predicted scores zero on all three
            p.append(0); r.append(0); ap.append(0)
            continue
        hits = tp[mine]
        tpc, fpc = hits.cumsum(), (1 - hits).cumsum()
        recall, precision = tpc / (n_gt + 1e-16), tpc / (tpc + fpc)
        r.append(recall[-1]); p.append(precision[-1]); ap.append(compute_ap(recall, precision))
    p, r, ap = np.array(p), np.array(r), np.array(ap)
    f1 = 2 * p * r / (p + r + 1e-16)
    return np.mean(p), np.mean(r), np.mean(ap), np.mean(f1)


def evaluation(val_dataloader, cfg, model, device, conf_thres=0.01, nms_thresh=0.4, iou_thres=0.5):
    """utils/utils.py:360-397 with the same signature and return value (``ap_per_class``'s 4-tuple, or None when the
    loader is empty).  Per batch: the pre-process (`float()/255`, or none for a uint8 (B,H,W,3) batch), ONE fused
    forward+decode+NMS call and ONE matching launch; detections, true-positive flags and counts stay on the GPU and
    come back in a single copy after the last batch.  ``model`` is a yolo_fastestv2_amd.Detector."""
    import numpy as np
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("evaluation: no CPU path, pass the MI355X device")
    labels, kept = [], []
    engines = set()
    scale = None
    for imgs, targets in val_dataloader:
        imgs = imgs.to(device)
        u8_hwc = imgs.dtype == torch.uint8 and imgs.dim() == 4 and imgs.shape[-1] == 3
        x = imgs if u8_hwc else imgs.float() / 255.0
        targets = targets.to(device).clone()
        labels += targets[:, 1].tolist()
        # normalised (cx, cy, w, h) -> corner pixels, in fp32 on the device like the reference (:372-376)
        c = targets[:, 2:].clone()
        targets[:, 2] = c[:, 0] - c[:, 2] / 2
        targets[:, 3] = c[:, 1] - c[:, 3] / 2
        targets[:, 4] = c[:, 0] + c[:, 2] / 2
        targets[:, 5] = c[:, 1] + c[:, 3] / 2
        if scale is None:
            scale = torch.tensor([cfg["width"], cfg["height"], cfg["width"], cfg["height"]]).to(device)
        targets[:, 2:] *= scale
        eng = model.engine_for(x)
        eng.set_anchors(cfg["anchors"])
        dets, _, cnt = eng.detect(x, conf_thres, nms_thresh)
        tp = eng.batch_statistics(dets, cnt, targets, iou_thres, sync=False)   # enqueue only: nothing waits until the last line
        engines.add(eng)
        live = torch.arange(dets.shape[1], device=device)[None, :] < cnt[:, None]      # image-major, rank order: the order
        kept.append((tp[live], dets[..., 4][live], dets[..., 5][live]))                 # sample_metrics is concatenated in
    # the sticky overflow word of EVERY engine used is read (and thereby cleared) before any return path, so that a flag set
    # here can never surface in a later, unrelated evaluation on the cached handle
    over = [eng.stats_overflowed() for eng in engines]
    bad = [eng.nonfinite() for eng in engines]
    if any(over):
        raise RuntimeError("evaluation: an image has more than 1024 targets (yfv2_batch_statistics limit)")
    if any(bad):
        raise RuntimeError("evaluation: an activation left the range of the default (fp16x3) plan (include/yfv2.h yfv2_nonfinite); "
                           "run on the fp32-matrix plan (YFV2_BF6=0 in the environment of the Python layer / yfv2_plan.fp32_matrix = 1)")
    if not kept:
        print("---- No detections over whole validation set ----")
        return None
    tp = torch.cat([k[0] for k in kept]).cpu().numpy().astype(np.float64)
    conf = torch.cat([k[1] for k in kept]).cpu().numpy()
    cls = torch.cat([k[2] for k in kept]).cpu().numpy()
    return ap_per_class(tp, conf, cls, labels)


def _engine_for_rows(device, rows, classes):
    for h in range(32, 2049, 32):  # square inputs first (the reference only ever uses H == W)
        if 3 * ((h // 16) ** 2 + (h // 32) ** 2) == rows:
            return get_engine(device, h, h, classes, 3)
    for h in range(32, 2049, 32):
        for w in range(32, 2049, 32):
            if 3 * ((h // 16) * (w // 16) + (h // 32) * (w // 32)) == rows:
                return get_engine(device, h, w, classes, 3)
    raise ValueError("cannot infer the input size from %d decode rows" % rows)
