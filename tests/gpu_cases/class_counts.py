#!/usr/bin/env python3
"""Parity of a non-default class count, run in its OWN interpreter by tests/test_gpu_parity.py (a fault in here must not take
the GPU test run down with it): Detector surface + decode + NMS + fused detect against the oracle, a progress marker after
every stage.  usage: class_counts.py CLASSES [HEIGHT WIDTH]   (default 352 352; other sizes exercise the general plan, the
two-launch decode + NMS and, beyond 2048 decode rows, the four-keys-per-thread sort of the NMS kernel)"""
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
    H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (352, 352)
    rows_n = 3 * ((H // 16) * (W // 16) + (H // 32) * (W // 32))
    dev = torch.device("cuda:0")
    anchors = [12.64, 19.39, 37.88, 51.48, 55.71, 138.31, 126.91, 78.23, 131.57, 214.55, 279.92, 258.87]
    w = yfv2.random_state_dict(7, classes=classes)
    torch.manual_seed(4)
    x = torch.rand(3, 3, H, W)
    mark("oracle forward")
    ref = oracle.forward(w, x)
    ref64 = oracle.forward64(w, x)
    mark("detector")
    m = yfv2.Detector(classes, 3, True).to(dev)
    m.load_state_dict(w)
    m.eval()
    got = m(x.to(dev))
    torch.cuda.synchronize()
    mark("forward done")
    for g, r, r64 in zip(got, ref, ref64):
        # the noise-floor rule of tests/test_gpu_parity.py: the device against float64, held to 3x the reference arithmetic's own error
        assert tuple(g.shape) == tuple(r.shape)
        e_dev = float((g.cpu().double() - r64).abs().max())
        e_ref = max(float((r.double() - r64).abs().max()), 2.0 ** -22 * max(1.0, float(r64.abs().max())))
        assert e_dev <= 3.0 * e_ref, "logits: %g from float64, the reference's fp32 %g" % (e_dev, e_ref)
    mark("logits ok")
    cfg = {"height": H, "width": W, "anchor_num": 3, "anchors": anchors}
    dec = yfv2.handel_preds(got, cfg, dev)
    assert tuple(dec.shape) == (3, rows_n, 5 + classes)
    o_dec = oracle.decode([t.cpu() for t in got], anchors, H)
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
    # every row a candidate, many classes, clusters of overlapping boxes: the sort at its full width (rows_n keys), the greedy
    # walk over hundreds of suppressions - bit-exact rows and indices against the oracle
    g = torch.Generator().manual_seed(11)
    syn = dec[:2].clone()
    syn[..., 4] = 0.31 + 0.69 * torch.rand(2, rows_n, generator=g)
    probs = torch.rand(2, rows_n, classes, generator=g) * 0.01
    hot = torch.randint(0, classes, (2, rows_n), generator=g)
    probs.scatter_(2, hot[..., None], 0.97 + 0.02 * torch.rand(2, rows_n, 1, generator=g))
    syn[..., 5:] = probs
    centers = torch.rand(2, 40, 2, generator=g) * torch.tensor([float(W), float(H)])
    which = torch.randint(0, 40, (2, rows_n), generator=g)
    syn[..., 0:2] = torch.gather(centers, 1, which[..., None].expand(-1, -1, 2)) + 6.0 * torch.randn(2, rows_n, 2, generator=g)
    syn[..., 2:4] = 20.0 + 80.0 * torch.rand(2, rows_n, 2, generator=g)
    syn[1, ::7, 4] = 0.75                                   # exact score ties (conf = obj * cls differs only through cls)
    rows2, idx2 = yfv2.nms_with_indices(syn, 0.3, 0.4)
    o_rows2, o_idx2 = oracle.non_max_suppression(syn.numpy(), 0.3, 0.4)
    for b in range(2):
        assert np.array_equal(rows2[b].numpy().view(np.uint32), o_rows2[b].view(np.uint32)), "stress NMS rows differ, image %d" % b
        assert np.array_equal(np.asarray(idx2[b]), o_idx2[b]), "stress NMS indices differ, image %d" % b
    mark("stress nms ok: %d candidates -> %s kept" % (rows_n, [int(r.shape[0]) for r in rows2]))
    print("PARITY OK classes=%d" % classes, flush=True)


if __name__ == "__main__":
    main()
