#!/usr/bin/env python3
"""Parity of a non-default class count, run in its OWN interpreter by tests/test_gpu_parity.py (a fault in here must not take
the GPU test run down with it): Detector surface + decode + NMS + fused detect against the oracle, a progress marker after
every stage.  usage: class_counts.py CLASSES"""
import faulthandler
import os
import sys

faulthandler.enable()
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import yolo_fastestv2_amd as yfv2  # noqa: E402
from oracle import yfv2_oracle as oracle  # noqa: E402


def mark(msg):
    print("[class_counts] " + msg, flush=True)


def main():
    classes = int(sys.argv[1])
    dev = torch.device("cuda:0")
    anchors = [12.64, 19.39, 37.88, 51.48, 55.71, 138.31, 126.91, 78.23, 131.57, 214.55, 279.92, 258.87]
    w = yfv2.random_state_dict(7, classes=classes)
    torch.manual_seed(4)
    x = torch.rand(3, 3, 352, 352)
    mark("oracle forward")
    ref = oracle.forward(w, x)
    mark("detector")
    m = yfv2.Detector(classes, 3, True).to(dev)
    m.load_state_dict(w)
    m.eval()
    got = m(x.to(dev))
    torch.cuda.synchronize()
    mark("forward done")
    for g, r in zip(got, ref):
        assert tuple(g.shape) == tuple(r.shape)
        scale = max(1.0, float(r.abs().max()))
        err = float((g.cpu() - r).abs().max())
        assert err <= 1e-4 * scale, "logits: max abs err %g (scale %g)" % (err, scale)
    mark("logits ok")
    cfg = {"height": 352, "width": 352, "anchor_num": 3, "anchors": anchors}
    dec = yfv2.handel_preds(got, cfg, dev)
    assert tuple(dec.shape) == (3, 1815, 5 + classes)
    o_dec = oracle.decode([t.cpu() for t in got], anchors, 352)
    d = np.abs(dec.numpy().astype(np.float64) - o_dec.astype(np.float64))
    assert (d[..., :4] <= 1e-4 * np.maximum(1.0, np.abs(o_dec[..., :4]))).all(), "decoded boxes: worst %g" % d[..., :4].max()
    assert d[..., 4:].max() <= 1e-5, "decoded scores: worst %g" % d[..., 4:].max()
    mark("decode ok")
    rows, idx = yfv2.nms_with_indices(dec, 0.3, 0.4)
    o_rows, o_idx = oracle.non_max_suppression(dec.numpy(), 0.3, 0.4)
    for b in range(3):
        assert np.array_equal(rows[b].numpy().view(np.uint32), o_rows[b].view(np.uint32)), "NMS rows differ, image %d" % b
        assert np.array_equal(np.asarray(idx[b]), o_idx[b]), "NMS indices differ, image %d" % b
    mark("nms ok")
    eng = m.engine_for(x.to(dev))
    eng.set_anchors(anchors)
    dd, ii, cc = eng.detect(x.to(dev), 0.3, 0.4)
    for b in range(3):
        n = int(cc[b])
        assert n == rows[b].shape[0], "fused detect kept %d, three calls kept %d" % (n, rows[b].shape[0])
        assert np.array_equal(ii[b, :n].cpu().numpy().astype(np.int64), np.asarray(idx[b]).astype(np.int64))
    mark("detect ok")
    print("PARITY OK classes=%d" % classes, flush=True)


if __name__ == "__main__":
    main()
