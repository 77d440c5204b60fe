"""Training loss (first slice): utils/loss.py compute_loss and its gradient w.r.t. the logits.

golden_loss.npz holds what the REFERENCE's own compute_loss and autograd produced (tests/golden/make_golden.py loss,
run where /root/reference exists): CPU part - the oracle restatement reproduces it exactly; GPU part (-m gpu) - the
HIP kernels (yfv2_loss through the C ABI, and the `compute_loss` drop-in with .backward()) match it:
losses within 1e-5 relative, gradients within 1e-5 of the largest gradient of their map."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import yfv2_oracle as oracle

sys.path.insert(0, GOLDEN)
from make_golden import LOSS_CASES, loss_case_inputs  # noqa: E402  (pure numpy generator of the seeded inputs)


@pytest.fixture(scope="module")
def golden_loss():
    return dict(np.load(os.path.join(GOLDEN, "golden_loss.npz")))


def _ref_grads(z, ci, preds):
    out = []
    for k, p in enumerate(preds):
        if k % 3 == 1:
            out.append(z["grad%d_%d" % (ci, k)].reshape(p.shape))
        else:
            g = np.zeros(p.size, np.float32)
            g[z["grad%d_%d_idx" % (ci, k)]] = z["grad%d_%d_val" % (ci, k)]
            out.append(g.reshape(p.shape))
    return out


def test_oracle_loss_reproduces_the_reference_golden(golden_loss, cfg):
    assert [tuple(c) for c in golden_loss["cases"]] == [tuple(c) for c in LOSS_CASES]
    for ci, (classes, B, T, seed) in enumerate(LOSS_CASES):
        preds, t = loss_case_inputs(classes, B, T, seed)
        assert np.array_equal(t, golden_loss["targets%d" % ci])
        p = [torch.from_numpy(x.copy()).requires_grad_() for x in preds]
        out = oracle.compute_loss(p, torch.from_numpy(t), cfg["anchors"], classes)
        out[3].backward()
        assert np.array_equal(np.asarray([float(v) for v in out], np.float32), golden_loss["loss%d" % ci]), ci
        for k, (a, r) in enumerate(zip(p, _ref_grads(golden_loss, ci, preds))):
            g = a.grad.numpy() if a.grad is not None else np.zeros_like(r)
            assert np.array_equal(g, r), (ci, k)


def test_oracle_build_target_edge_cases(cfg):
    """No labels; a label whose box matches no anchor; a label in the corner cell (neighbour offsets must not leave the map)."""
    shapes = [(22, 22), (11, 11)]
    empty = oracle.build_target(shapes, np.zeros((0, 6), np.float32), cfg["anchors"])
    assert all(len(s[0]) == 0 for s in empty)
    far = oracle.build_target(shapes, np.asarray([[0, 1, 0.5, 0.5, 0.001, 0.001]], np.float32), cfg["anchors"])
    assert all(len(s[0]) == 0 for s in far)               # 0.02 x 0.02 cells: ratio to every anchor > 2
    corner = oracle.build_target(shapes, np.asarray([[0, 1, 0.999, 0.001, 0.1, 0.15]], np.float32), cfg["anchors"])
    assert len(corner[0][0]) == 1                         # stride 16: the corner cell only - both neighbour tests fail at the border (:104-105)
    inner = oracle.build_target(shapes, np.asarray([[0, 1, 0.52, 0.48, 0.1, 0.15]], np.float32), cfg["anchors"])
    assert len(inner[0][0]) == 3                          # centre + one horizontal + one vertical neighbour
    for (b, a, gj, gi, tb, an, c), (h, w) in zip(corner, shapes):
        if len(b):
            assert 0 <= int(gi.min()) and int(gi.max()) <= w - 1 and 0 <= int(gj.min()) and int(gj.max()) <= h - 1


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(LOSS_CASES)))
def test_hip_loss_and_gradients_vs_reference_golden(golden_loss, cfg, ci):
    import yolo_fastestv2_amd as yfv2
    classes, B, T, seed = LOSS_CASES[ci]
    preds, t = loss_case_inputs(classes, B, T, seed)
    dev = torch.device("cuda:0")
    eng = yfv2.get_engine(dev, 352, 352, classes, 3)
    eng.set_anchors(cfg["anchors"])
    dp = [torch.from_numpy(p).to(dev) for p in preds]
    losses, grads = eng.loss(dp, torch.from_numpy(t), want_grad=True)
    ref = golden_loss["loss%d" % ci]
    got = losses.cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-5 * max(1.0, float(np.abs(ref).max())), (got, ref)
    for k, (g, r) in enumerate(zip(grads, _ref_grads(golden_loss, ci, preds))):
        err = float(np.abs(g.cpu().numpy() - r).max())
        assert err <= 1e-5 * max(float(np.abs(r).max()), 1e-3), "case %d map %d: max abs err %g (largest gradient %g)" % (ci, k, err, np.abs(r).max())
    # forward only: same values, no gradient buffers touched
    l2, g2 = eng.loss(dp, torch.from_numpy(t))
    assert g2 is None and torch.equal(l2, losses)


@pytest.mark.gpu
def test_compute_loss_drop_in_with_backward(golden_loss, cfg):
    """train.py:105-108: compute_loss(preds, targets, cfg, device) -> 4-tuple; total_loss.backward() reaches the logits."""
    import yolo_fastestv2_amd as yfv2
    ci = 1
    classes, B, T, seed = LOSS_CASES[ci]
    preds, t = loss_case_inputs(classes, B, T, seed)
    dev = torch.device("cuda:0")
    dp = [torch.from_numpy(p).to(dev).requires_grad_() for p in preds]
    c = dict(cfg, classes=classes)
    iou_loss, obj_loss, cls_loss, total_loss = yfv2.compute_loss(dp, torch.from_numpy(t).to(dev), c, dev)
    for v in (iou_loss, obj_loss, cls_loss, total_loss):
        assert tuple(v.shape) == (1,) and v.device.type == "cuda"
    ref = golden_loss["loss%d" % ci]
    assert abs(float(total_loss) - ref[3]) <= 1e-5 * ref[3] and abs(float(iou_loss) - ref[0]) <= 1e-5 * max(1, ref[0])
    (total_loss * 2.0).backward()                        # a scaled loss scales the gradients (subdivisions, AMP-style scaling)
    for k, (p, r) in enumerate(zip(dp, _ref_grads(golden_loss, ci, preds))):
        assert float((p.grad.cpu() - 2.0 * torch.from_numpy(r)).abs().max()) <= 2e-5 * max(float(np.abs(r).max()), 1e-3), k
