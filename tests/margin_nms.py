"""Test helper: which survivor sets of `non_max_suppression` (utils/utils.py:232-296) are reachable from a decoded tensor
when every score may move by the tolerance the parity tests grant (obj / cls 1e-5, boxes 1e-4 relative)?

Two valid fp32 executions of the reference forward differ by ~1e-6 on scores, so an end-to-end comparison
of survivor INDICES can only demand identity where no decision of the greedy walk sits on a numerical margin.  The rule is
interval logic over the oracle's decoded rows, cascades included:

  presence   obj and conf = max_j obj*cls_j are compared with conf_thres: both clearly above -> present, one clearly below ->
             absent, else uncertain
  order      candidates are visited by descending conf; two whose conf differ by <= EPS_TIE may be visited in either order
  IoU        of the class-offset fp32 boxes (utils.py:283-285; the offset quantises them, see classify): > thr + eps suppresses,
             < thr - eps does not, else either; eps = EPS_IOU + what a one-bucket move of the quantised edges does to the pair
  status     walking down the order: SUPPRESSED if a certainly-earlier, certainly-KEPT candidate certainly suppresses it;
             KEPT if present and no earlier-or-tied candidate that is KEPT or UNCERTAIN can suppress it; else UNCERTAIN
  max_det    (utils.py:287-288) a KEPT candidate must be reported if fewer than 300 KEPT-or-UNCERTAIN precede it, a candidate
             may be reported only if it is KEPT or UNCERTAIN and fewer than 300 KEPT precede it

`check` applies the rule as a CERTIFICATE relative to the survivor list under test (see its docstring): interval logic that
propagates "uncertain" down the greedy walk (`classify`, kept for the soundness tests) cascades in dense scenes - with 1815
mutually overlapping candidates one open decision makes hundreds of rows uncertain and the check says nothing.
The three margins default to what the COCO-weights tests grant; a caller whose two executions agree less tightly (random
weights: larger logits, larger absolute differences) passes the margins it MEASURED (`eps_conf`, `eps_tie`, `eps_iou`).
`check(...)` returns the rows a device result must contain, the rows it may contain, and the number of UNCERTAIN rows (the
"margin count" the parity records quote).  Test infrastructure only - nothing in the product imports this.
"""
import numpy as np

EPS_CONF = 1e-4   # distance of obj / conf from conf_thres below which either side is accepted (tests/test_gpu_parity.py)
EPS_TIE = 2e-5    # conf difference below which the visiting order of two candidates is open (scores agree to 1e-5 each)
EPS_IOU = 1e-4    # added on top of the IoU range the edge uncertainties give (check) / flat IoU margin (classify: 20x this)
BOX_RTOL = 1e-4   # decoded cx, cy, w, h agree to 1e-4 * max(1, |value|) between two valid executions (tests/test_gpu_parity.py)
MAX_DET = 300
MAX_WH = 4096.0

KEPT, UNCERTAIN, SUPPRESSED = 0, 1, 2


def classify(dec_img, conf_thres, iou_thres, classes=None, eps_conf=None, eps_tie=None, eps_iou=None):
    """dec_img: (rows, 5 + nc) decoded rows of ONE image (the oracle's).  Returns (row ids in visiting order, status per
    visited row, conf per visited row)."""
    EPS_CONF, EPS_TIE, EPS_IOU = (globals()["EPS_CONF"] if eps_conf is None else eps_conf, globals()["EPS_TIE"] if eps_tie is None else eps_tie,
                                  20 * globals()["EPS_IOU"] if eps_iou is None else eps_iou)
    d = np.asarray(dec_img, dtype=np.float32)
    obj = d[:, 4]
    pool = np.flatnonzero(obj > conf_thres - EPS_CONF)
    if pool.size == 0:
        return pool, np.zeros(0, np.int8), np.zeros(0, np.float32)
    x = d[pool]
    p = x[:, 5:] * x[:, 4:5]
    cls = p.argmax(1)
    conf = p[np.arange(p.shape[0]), cls]
    keep = conf > conf_thres - EPS_CONF
    if classes is not None:
        keep &= np.isin(cls, np.asarray(classes))
    pool, x, p, cls, conf = pool[keep], x[keep], p[keep], cls[keep], conf[keep]
    n = pool.size
    if n == 0:
        return pool, np.zeros(0, np.int8), np.zeros(0, np.float32)
    present = (x[:, 4] > conf_thres + EPS_CONF) & (conf > conf_thres + EPS_CONF)
    # a second class within EPS_TIE of the best one: the argmax (and with it the class offset) is open
    if p.shape[1] > 1:
        second = np.partition(p, -2, axis=1)[:, -2]
        cls_open = (conf - second) <= EPS_TIE
    else:
        cls_open = np.zeros(n, bool)
    # boxes as the reference's NMS sees them (utils.py:283-285): xyxy in fp32 PLUS the class offset cls * 4096 in fp32 - at
    # class 36 that is a coordinate near 147 456, where fp32 is spaced 1/64 pixel apart: the offset QUANTISES the box, and a
    # 1e-4 relative difference between two executions' coordinates can land in the neighbouring bucket.  So IoU is computed
    # on the quantised boxes, and every pair's margin is widened by what a one-bucket move of each edge can do to its IoU
    half_w, half_h = x[:, 2] / np.float32(2), x[:, 3] / np.float32(2)
    box32 = np.stack((x[:, 0] - half_w, x[:, 1] - half_h, x[:, 0] + half_w, x[:, 1] + half_h), 1).astype(np.float32)
    box32 = box32 + (cls.astype(np.float32) * np.float32(MAX_WH))[:, None]
    raw = box32.astype(np.float64)
    area = (raw[:, 2] - raw[:, 0]) * (raw[:, 3] - raw[:, 1])
    bucket = np.spacing(np.abs(box32).max(1).astype(np.float32)).astype(np.float64)                 # fp32 spacing at the offset coordinate
    side = np.maximum(np.minimum(raw[:, 2] - raw[:, 0], raw[:, 3] - raw[:, 1]), 1e-3)
    quant = 4.0 * bucket / side                                                                          # IoU change of a one-bucket move of the edges
    order = np.argsort(-conf, kind="stable")
    pool, present, cls, conf, cls_open, raw, area, quant = (pool[order], present[order], cls[order], conf[order], cls_open[order], raw[order],
                                                            area[order], quant[order])
    status = np.full(n, KEPT, np.int8)
    for i in range(n):
        # candidates that may be visited before i: everything above it in the order, plus those below it whose conf is tied
        hi = i + 1
        while hi < n and conf[i] - conf[hi] <= EPS_TIE:
            hi += 1
        js = np.arange(hi)
        js = js[js != i]
        if js.size == 0:
            status[i] = KEPT if present[i] else UNCERTAIN
            continue
        iw = np.clip(np.minimum(raw[js, 2], raw[i, 2]) - np.maximum(raw[js, 0], raw[i, 0]), 0, None)
        ih = np.clip(np.minimum(raw[js, 3], raw[i, 3]) - np.maximum(raw[js, 1], raw[i, 1]), 0, None)
        inter = iw * ih
        iou = inter / np.maximum(area[js] + area[i] - inter, 1e-30)
        same = (cls[js] == cls[i])
        # an open argmax moves the class offset: such a pair's IoU on the quantised boxes says nothing, compare un-offset
        open_pair = cls_open[js] | cls_open[i]
        if open_pair.any():
            dj = (cls[js] - cls[i]).astype(np.float64) * MAX_WH
            iw2 = np.clip(np.minimum(raw[js, 2] - dj, raw[i, 2]) - np.maximum(raw[js, 0] - dj, raw[i, 0]), 0, None)
            ih2 = np.clip(np.minimum(raw[js, 3] - dj, raw[i, 3]) - np.maximum(raw[js, 1] - dj, raw[i, 1]), 0, None)
            iou = np.where(open_pair, iw2 * ih2 / np.maximum(area[js] + area[i] - iw2 * ih2, 1e-30), iou)
        maybe_same = same | open_pair
        eps_pair = np.minimum(EPS_IOU + quant[js] + quant[i], 0.5)
        certainly_before = (js < i) & (conf[js] - conf[i] > EPS_TIE)
        sure = certainly_before & same & ~open_pair & (iou > iou_thres + eps_pair) & (status[js] == KEPT)
        if sure.any():
            status[i] = SUPPRESSED
            continue
        # rows below i in the order have no status yet: a tied one counts as a possible suppressor unless it is absent
        st_j = np.where(js < i, status[js], UNCERTAIN)
        threat = maybe_same & (iou > iou_thres - eps_pair) & (st_j != SUPPRESSED)
        status[i] = UNCERTAIN if (threat.any() or not present[i] or cls_open[i]) else KEPT
    return pool, status, conf


def _candidates(dec_img, conf_thres, classes, EPS_CONF, EPS_TIE, box_rtol=None):
    """rows that may take part (obj, conf within EPS_CONF of passing): ids, present (clearly passing), cls, conf, cls_open,
    fp32 offset boxes as float64, area, per-row quantisation term"""
    d = np.asarray(dec_img, dtype=np.float32)
    pool = np.flatnonzero(d[:, 4] > conf_thres - EPS_CONF)
    x = d[pool]
    p = x[:, 5:] * x[:, 4:5]
    cls = p.argmax(1) if pool.size else np.zeros(0, np.int64)
    conf = p[np.arange(p.shape[0]), cls] if pool.size else np.zeros(0, np.float32)
    keep = conf > conf_thres - EPS_CONF
    if classes is not None:
        keep &= np.isin(cls, np.asarray(classes))
    pool, x, p, cls, conf = pool[keep], x[keep], p[keep], cls[keep], conf[keep]
    present = (x[:, 4] > conf_thres + EPS_CONF) & (conf > conf_thres + EPS_CONF)
    if p.shape[0] and p.shape[1] > 1:
        cls_open = (conf - np.partition(p, -2, axis=1)[:, -2]) <= EPS_TIE
    else:
        cls_open = np.zeros(pool.size, bool)
    half_w, half_h = x[:, 2] / np.float32(2), x[:, 3] / np.float32(2)
    box32 = np.stack((x[:, 0] - half_w, x[:, 1] - half_h, x[:, 0] + half_w, x[:, 1] + half_h), 1).astype(np.float32)
    box32 = box32 + (cls.astype(np.float32) * np.float32(MAX_WH))[:, None]
    raw = box32.astype(np.float64)
    area = (raw[:, 2] - raw[:, 0]) * (raw[:, 3] - raw[:, 1])
    # how far an edge of this box may sit from where the other execution has it: one fp32 bucket at the offset coordinate
    # (the class offset quantises: 1/64 pixel at class 36) plus the decode tolerance the parity tests grant on cx, cy, w, h
    bucket = np.spacing(np.abs(box32).max(1).astype(np.float32)).astype(np.float64) if pool.size else np.zeros(0)
    delta = bucket + (BOX_RTOL * 1.5 if box_rtol is None else box_rtol) * np.maximum(1.0, np.abs(x[:, :4]).max(1).astype(np.float64))
    return pool, present, cls, conf, cls_open, raw, area, delta


def check(dec_img, got_rows, conf_thres, iou_thres, classes=None, eps_conf=None, eps_tie=None, eps_iou=None, box_rtol=None):
    """The CERTIFICATE form of the rule, relative to the survivor list `got_rows` a device reported for this image (in its
    reported order) - no cascades, because every decision is checked against the device's own kept set K:
      * a kept row must be able to pass the thresholds, and no kept row that certainly precedes it may certainly suppress it
        (IoU > thr + eps on the quantised class-offset boxes)                                                 -> else `forbidden`
      * a candidate that clearly passes the thresholds and is NOT kept must have a possible suppressor in K (a kept row that
        may precede it with IoU > thr - eps) - unless it ranks behind the 300th kept row (max_det)            -> else `missing`
    A greedy NMS result is exactly the set with these two properties, so a valid execution within the margins always passes
    and anything else is a wrong result.  `n_uncertain` counts the decisions that only hold thanks to a margin (a kept row's
    nearest certainly-preceding kept neighbour within eps of the threshold, a dropped row whose best suppressor is within eps,
    a row within EPS_CONF of conf_thres).  `box_rtol`: how far (relative to max(1, |coordinate|)) a decoded cx, cy, w, h of the
    execution under test may sit from `dec_img`'s (default 1.5 * BOX_RTOL)."""
    EPS_C = EPS_CONF if eps_conf is None else eps_conf
    EPS_T = EPS_TIE if eps_tie is None else eps_tie
    EPS_I = EPS_IOU if eps_iou is None else eps_iou
    pool, present, cls, conf, cls_open, raw, area, delta = _candidates(dec_img, conf_thres, classes, EPS_C, EPS_T, box_rtol)
    pos = {int(r): i for i, r in enumerate(pool.tolist())}
    got = [int(v) for v in got_rows]
    forbidden = [r for r in got if r not in pos]          # clearly below a threshold (or filtered class)
    K = np.asarray([pos[r] for r in got if r in pos], dtype=np.int64)
    n_margin = int((~present[K]).sum()) if K.size else 0
    missing = []

    def iou_bounds(i, js):
        """(lowest, highest) IoU the pair can have when every edge may move by its box's delta (+- EPS_I on top)"""
        dj = np.where(cls_open[js] | cls_open[i], (cls[js] - cls[i]).astype(np.float64) * MAX_WH, 0.0)   # open argmax: compare un-offset
        iw = np.minimum(raw[js, 2] - dj, raw[i, 2]) - np.maximum(raw[js, 0] - dj, raw[i, 0])
        ih = np.minimum(raw[js, 3] - dj, raw[i, 3]) - np.maximum(raw[js, 1] - dj, raw[i, 1])
        d2 = delta[js] + delta[i]
        wj, hj, wi, hi_ = raw[js, 2] - raw[js, 0], raw[js, 3] - raw[js, 1], raw[i, 2] - raw[i, 0], raw[i, 3] - raw[i, 1]

        def iou_of(iw_, ih_, aj, ai):
            inter = np.clip(iw_, 0, None) * np.clip(ih_, 0, None)
            return inter / np.maximum(aj + ai - inter, 1e-30)
        lo = iou_of(iw - d2, ih - d2, (wj + 2 * delta[js]) * (hj + 2 * delta[js]), (wi + 2 * delta[i]) * (hi_ + 2 * delta[i])) - EPS_I
        hi = iou_of(np.minimum(iw + d2, np.minimum(wj, wi) + d2), np.minimum(ih + d2, np.minimum(hj, hi_) + d2),
                    np.clip(wj - 2 * delta[js], 0, None) * np.clip(hj - 2 * delta[js], 0, None),
                    np.clip(wi - 2 * delta[i], 0, None) * np.clip(hi_ - 2 * delta[i], 0, None)) + EPS_I
        same = (cls[js] == cls[i]) | cls_open[js] | cls_open[i]
        return np.where(same, lo, 0.0), np.where(same, np.minimum(hi, 1.0 + EPS_I), 0.0)

    if K.size:
        for n, i in enumerate(K.tolist()):
            js = K[:n][conf[K[:n]] - conf[i] > EPS_T]      # kept rows that certainly precede i
            if js.size:
                lo, hi = iou_bounds(i, js)
                sure_same = (cls[js] == cls[i]) & ~cls_open[js] & ~cls_open[i]
                if (sure_same & (lo > iou_thres)).any():
                    forbidden.append(int(pool[i]))
                elif (hi > iou_thres).any():
                    n_margin += 1
        cut = conf[K[-1]] if K.size >= MAX_DET else -np.inf   # behind the 300th kept row nothing has to be reported
        kept_set = set(K.tolist())
        for i in np.flatnonzero(present).tolist():
            if i in kept_set or conf[i] < cut + EPS_T:
                continue
            js = K[conf[K] - conf[i] > -EPS_T]              # kept rows that may precede i
            lo, hi = iou_bounds(i, js) if js.size else (np.zeros(0), np.zeros(0))
            if not (hi > iou_thres).any():
                missing.append(int(pool[i]))
            elif not (lo > iou_thres).any():
                n_margin += 1
    else:
        missing = [int(pool[i]) for i in np.flatnonzero(present).tolist()]
    return {"missing": sorted(missing), "forbidden": sorted(forbidden), "n_uncertain": n_margin, "n_candidates": int(pool.size),
            "n_must": int(present.sum()), "n_may": int(pool.size)}
