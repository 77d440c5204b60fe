"""The margin rule the end-to-end GPU tests use (tests/margin_nms.py) is sound: the reference's own survivors (golden
vectors made by /root/reference's non_max_suppression) always satisfy it, so do the oracle's survivors of copies of the
decoded tensor perturbed within the parity tolerances, and a survivor set with a clearly wrong row is rejected."""
import numpy as np

import margin_nms
from conftest import unpack_ragged
from oracle import yfv2_oracle as oracle


def _perturb(dec, rng):
    d = dec.astype(np.float64).copy()
    d[..., :4] += rng.uniform(-1, 1, d[..., :4].shape) * 1e-4 * np.maximum(1.0, np.abs(d[..., :4]))
    d[..., 4:] += rng.uniform(-1, 1, d[..., 4:].shape) * 1e-5
    return np.clip(d, 0, None).astype(np.float32)


def test_reference_survivors_satisfy_the_margin_rule(golden_real, golden_rand, golden_stress):
    for z, cases in ((golden_real, (("nms_03_04", 0.3, 0.4), ("nms_001_04", 0.01, 0.4), ("nms_03_045", 0.3, 0.45))),
                     (golden_rand, (("nms_03_04", 0.3, 0.4), ("nms_001_04", 0.01, 0.4))),
                     (golden_stress, (("nms_03_04", 0.3, 0.4), ("nms_001_04", 0.01, 0.4), ("nms_025_06", 0.25, 0.6)))):
        for prefix, conf, iou in cases:
            _, g_idx = unpack_ragged(z, prefix)
            for b in range(len(g_idx)):
                r = margin_nms.check(z["decoded"][b], g_idx[b], conf, iou)
                assert not r["missing"] and not r["forbidden"], (prefix, b, r)


def test_perturbed_executions_satisfy_the_margin_rule(golden_real, golden_stress):
    rng = np.random.default_rng(7)
    for z, conf, iou in ((golden_real, 0.01, 0.4), (golden_stress, 0.3, 0.4), (golden_stress, 0.25, 0.6)):
        dec = z["decoded"]
        for trial in range(3):
            _, idx = oracle.non_max_suppression(_perturb(dec, rng), conf, iou)
            for b in range(dec.shape[0]):
                r = margin_nms.check(dec[b], idx[b], conf, iou)
                assert not r["missing"] and not r["forbidden"], (conf, iou, trial, b, r)


def test_wrong_survivors_are_rejected(golden_real):
    dec = golden_real["decoded"]
    _, g_idx = unpack_ragged(golden_real, "nms_03_04")
    b = int(np.argmax([len(i) for i in g_idx]))
    good = [int(v) for v in g_idx[b]]
    assert len(good) >= 2
    r = margin_nms.check(dec[b], good[1:], 0.3, 0.4)          # the best detection dropped
    assert good[0] in r["missing"]
    low = int(np.argmin(dec[b][:, 4]))                         # a row far below the threshold reported
    r = margin_nms.check(dec[b], good + [low], 0.3, 0.4)
    assert low in r["forbidden"]
    # a clearly suppressed row reported: the candidate that overlaps the winner most among the non-survivors
    pool, status, _ = margin_nms.classify(dec[b], 0.3, 0.4)
    sup = [int(p) for p, s in zip(pool, status) if s == margin_nms.SUPPRESSED]
    if sup:
        r = margin_nms.check(dec[b], good + sup[:1], 0.3, 0.4)
        assert sup[0] in r["forbidden"]
