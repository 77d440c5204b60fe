"""Test helper: which survivor sets of `non_max_suppression` (utils/utils.py:232-296) are reachable from a decoded tensor
when every score may move by the tolerance the parity tests grant (obj / cls 1e-5, boxes 1e-4 relative)?

Two valid fp32 executions of the reference forward differ by ~1e-6 on scores, so an end-to-end comparison
of survivor INDICES can only demand identity where no decision of the greedy walk sits on a numerical margin.  The rule is
interval logic over the oracle's decoded rows, cascades included:

  presence   obj and conf = max_j obj*cls_j are compared with conf_thres: both clearly above -> present, one clearly below ->
             absent, else uncertain
  order      candidates are visited by descending conf; two whose conf differ by <= EPS_TIE may be visited in either order
  IoU        of the class-offset boxes (utils.py:283-285): > thr + EPS_IOU suppresses, < thr - EPS_IOU does not, else either
  status     walking down the order: SUPPRESSED if a certainly-earlier, certainly-KEPT candidate certainly suppresses it;
             KEPT if present and no earlier-or-tied candidate that is KEPT or UNCERTAIN can suppress it; else UNCERTAIN
  max_det    (utils.py:287-288) a KEPT candidate must be reported if fewer than 300 KEPT-or-UNCERTAIN precede it, a candidate
             may be reported only if it is KEPT or UNCERTAIN and fewer than 300 KEPT precede it

The three margins default to what the COCO-weights tests grant; a caller whose two executions agree less tightly (random
weights: larger logits, larger absolute differences) passes the margins it MEASURED (`eps_conf`, `eps_tie`, `eps_iou`).
`check(...)` returns the rows a device result must contain, the rows it may contain, and the number of UNCERTAIN rows (the
"margin count" the parity records quote).  Test infrastructure only - nothing in the product imports this.
"""
import numpy as np

EPS_CONF = 1e-4   # distance of obj / conf from conf_thres below which either side is accepted (tests/test_gpu_parity.py)
EPS_TIE = 2e-5    # conf difference below which the visiting order of two candidates is open (scores agree to 1e-5 each)
EPS_IOU = 2e-3    # IoU distance from iou_thres below which either decision is accepted (boxes agree to 1e-4 relative)
MAX_DET = 300
MAX_WH = 4096.0

KEPT, UNCERTAIN, SUPPRESSED = 0, 1, 2


def classify(dec_img, conf_thres, iou_thres, classes=None, eps_conf=None, eps_tie=None, eps_iou=None):
    """dec_img: (rows, 5 + nc) decoded rows of ONE image (the oracle's).  Returns (row ids in visiting order, status per
    visited row, conf per visited row)."""
    EPS_CONF, EPS_TIE, EPS_IOU = (globals()["EPS_CONF"] if eps_conf is None else eps_conf, globals()["EPS_TIE"] if eps_tie is None else eps_tie,
                                  globals()["EPS_IOU"] if eps_iou is None else eps_iou)
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
    half_w, half_h = x[:, 2] / np.float32(2), x[:, 3] / np.float32(2)
    raw = np.stack((x[:, 0] - half_w, x[:, 1] - half_h, x[:, 0] + half_w, x[:, 1] + half_h), 1).astype(np.float64)
    area = (raw[:, 2] - raw[:, 0]) * (raw[:, 3] - raw[:, 1])
    order = np.argsort(-conf, kind="stable")
    pool, present, cls, conf, cls_open, raw, area = pool[order], present[order], cls[order], conf[order], cls_open[order], raw[order], area[order]
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
        maybe_same = same | cls_open[js] | cls_open[i]
        certainly_before = (js < i) & (conf[js] - conf[i] > EPS_TIE)
        sure = certainly_before & same & ~cls_open[js] & ~cls_open[i] & (iou > iou_thres + EPS_IOU) & (status[js] == KEPT)
        if sure.any():
            status[i] = SUPPRESSED
            continue
        # rows below i in the order have no status yet: a tied one counts as a possible suppressor unless it is absent
        st_j = np.where(js < i, status[js], UNCERTAIN)
        threat = maybe_same & (iou > iou_thres - EPS_IOU) & (st_j != SUPPRESSED)
        status[i] = UNCERTAIN if (threat.any() or not present[i] or cls_open[i]) else KEPT
    return pool, status, conf


def check(dec_img, got_rows, conf_thres, iou_thres, classes=None, **eps):
    """got_rows: the survivor indices a device reported for this image.  Returns a dict with the rows it must / may
    contain violated (`missing`, `forbidden`), and the margin counts."""
    pool, status, _ = classify(dec_img, conf_thres, iou_thres, classes, **eps)
    must, may = set(), set()
    n_kept = n_possible = 0
    for r, st in zip(pool.tolist(), status.tolist()):
        if st == KEPT:
            if n_possible < MAX_DET:
                must.add(r)
            if n_kept < MAX_DET:
                may.add(r)
            n_kept += 1
            n_possible += 1
        elif st == UNCERTAIN:
            if n_kept < MAX_DET:
                may.add(r)
            n_possible += 1
    got = set(int(v) for v in got_rows)
    return {"missing": sorted(must - got), "forbidden": sorted(got - may), "n_must": len(must), "n_may": len(may),
            "n_uncertain": int((status == UNCERTAIN).sum()), "n_candidates": int(pool.size)}
