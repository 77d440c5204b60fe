"""SURVEY.md 8(f) row 3, the checker of the training path: the ORACLE's restatement of one training
iteration (train.py:93-112 - train-mode forward with batch-statistics BatchNorm, compute_loss, backward, one
SGD(momentum 0.949, weight_decay 0.0005) step) against tests/golden/golden_train.npz, which tests/golden/make_golden.py
`train` produced by running the reference's OWN modules (model.detector.Detector in train(), utils.loss.compute_loss,
torch.optim.SGD) on the same seeded weights, images and labels.  These are the parity targets of the train-mode forward /
backward / optimizer kernels (tests/test_train_gpu.py); nothing here touches the product path."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import yfv2_oracle as oracle

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden  # noqa: E402  (only its seeded-input helper and case list: nothing of the reference is imported)


@pytest.fixture(scope="module")
def golden_train():
    return np.load(os.path.join(GOLDEN, "golden_train.npz"))


@pytest.mark.parametrize("ci", range(len(make_golden.TRAIN_CASES)))
def test_train_step_oracle_matches_reference_golden(golden_train, ci):
    g = golden_train
    classes, B, T, seed, lr = make_golden.TRAIN_CASES[ci]
    assert tuple(g["cases"][ci]) == (classes, B, T, seed) and float(g["lr"][ci]) == lr
    w, x, t = make_golden.train_case_inputs(classes, B, T, seed)
    assert np.array_equal(t, g["targets%d" % ci])
    anchors = [float(a) for a in np.load(os.path.join(GOLDEN, "cfg_coco.npz"))["anchors"]]
    r = oracle.train_step(w, torch.from_numpy(x), torch.from_numpy(t), anchors, classes, lr)
    for a, b in zip(g["loss%d" % ci], r["losses"]):
        assert abs(float(a) - b) <= 1e-6 * max(1.0, abs(float(a)))
    names = [str(n) for n in g["names%d" % ci]]
    assert sorted(r["grads"]) == names                                   # every parameter of the reference module has a gradient
    for k, gn, gm in zip(names, g["gnorm%d" % ci], g["gmax%d" % ci]):
        assert abs(float(r["grads"][k].double().norm()) - gn) <= 1e-5 * max(gn, 1e-9), k
        assert abs(float(r["grads"][k].abs().max()) - gm) <= 1e-5 * max(gm, 1e-9), k
    for key in g.files:
        if key.startswith("grad%d:" % ci):
            k = key.split(":", 1)[1]
            ref = g[key]
            assert np.abs(r["grads"][k].numpy() - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-9), k
        elif key.startswith("after%d:" % ci):
            k = key.split(":", 1)[1]
            ref = g[key]
            got = r["new_w"][k].numpy()
            assert got.dtype == ref.dtype and np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() <= 1e-6 * max(1.0, np.abs(ref).max()), k
    for pi in range(6):
        ref = g["pred%d_%d" % (ci, pi)]
        assert np.abs(r["preds"][pi][0].numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_train_mode_batchnorm_differs_from_eval_and_tracks_statistics():
    """sanity of the restatement itself: train-mode logits are not the eval-mode ones, running statistics move by 0.1 of
    the way towards the batch statistics, num_batches_tracked counts"""
    classes, B, T, seed, lr = make_golden.TRAIN_CASES[1]
    w, x, t = make_golden.train_case_inputs(classes, B, T, seed)
    xt = torch.from_numpy(x)
    ev = oracle.forward(w, xt)
    with torch.no_grad():
        tr, bn = oracle.train_forward(w, xt)
    assert max(float((a - b).abs().max()) for a, b in zip(ev, tr)) > 1e-3
    name = "backbone.first_conv.1"
    y = torch.nn.functional.conv2d(xt, w["backbone.first_conv.0.weight"], None, 2, 1)
    mean, var = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=True)
    assert torch.allclose(bn[name + ".running_mean"], 0.9 * w[name + ".running_mean"] + 0.1 * mean, atol=1e-6)
    assert torch.allclose(bn[name + ".running_var"], 0.9 * w[name + ".running_var"] + 0.1 * var, atol=1e-6)
    anchors = [float(a) for a in np.load(os.path.join(GOLDEN, "cfg_coco.npz"))["anchors"]]
    r = oracle.train_step(w, xt, torch.from_numpy(t), anchors, classes, lr)
    assert int(r["new_w"][name + ".num_batches_tracked"]) == int(w[name + ".num_batches_tracked"]) + 1
    # a second step re-uses the momentum buffers: buf = 0.949 * buf + (grad + 0.0005 * w)
    r2 = oracle.train_step(r["new_w"], xt, torch.from_numpy(t), anchors, classes, lr, momentum_buf=r["momentum_buf"])
    k = "output_cls_layers.bias"
    want = r["momentum_buf"][k] * 0.949 + (r2["grads"][k] + 0.0005 * r["new_w"][k])
    assert torch.allclose(r2["momentum_buf"][k], want, atol=1e-7)
    assert r2["losses"][3] < r["losses"][3] * 1.5          # and nothing blew up


def test_warmup_schedule():
    """train.py:101-106: lr = base * (batch_num / (5 * len(loader))) ** 4 during the first five epochs' worth of batches"""
    assert oracle.warmup_lr(0.001, 0, 100) == 0.0
    assert oracle.warmup_lr(0.001, 250, 100) == 0.001 * (250 / 500) ** 4
    assert oracle.warmup_lr(0.001, 500, 100) == 0.001
    assert oracle.warmup_lr(0.001, 501, 100) == 0.001


def test_relu_decision_replay_and_why_it_is_needed():
    """train_step(relu_decisions=...) - what tests/test_train_gpu.py compares gradients on.  (a) replaying a run on its own
    decisions reproduces it exactly; (b) the reference's arithmetic itself (fp32) takes a few of the ~1e7 decisions
    differently from a float64 evaluation, every one of them on a pre-activation within 1e-3 of zero, and (c) that alone moves
    gradients by far more than fp32 rounding does: on the SAME decisions fp32 and float64 agree an order of magnitude better."""
    classes, B, T, seed, lr = make_golden.TRAIN_CASES[0]
    w, x, t = make_golden.train_case_inputs(classes, B, T, seed)
    anchors = [float(a) for a in np.load(os.path.join(GOLDEN, "cfg_coco.npz"))["anchors"]]
    xt, tt = torch.from_numpy(x), torch.from_numpy(t)
    r32 = oracle.train_step(w, xt, tt, anchors, classes, lr)
    assert len(r32["pre_relu"]) == 46                                     # stem, 35 in the backbone's blocks, 2 + 8 in the FPN
    dec32 = {k: v > 0 for k, v in r32["pre_relu"].items()}
    again = oracle.train_step(w, xt, tt, anchors, classes, lr, relu_decisions=dec32)
    assert all(torch.equal(again["grads"][k], r32["grads"][k]) for k in r32["grads"]) and again["losses"] == r32["losses"]
    w64 = {k: (v.double() if v.is_floating_point() else v) for k, v in w.items()}
    free64 = oracle.train_step(w64, xt.double(), tt, anchors, classes, lr)
    flips, worst = 0, 0.0
    for k, pre in free64["pre_relu"].items():
        d = dec32[k] != (pre > 0)
        flips += int(d.sum())
        if d.any():
            worst = max(worst, float(pre[d].abs().max()))
    assert 0 < flips < 50 and worst < 1e-3, (flips, worst)
    same64 = oracle.train_step(w64, xt.double(), tt, anchors, classes, lr, relu_decisions=dec32)

    def rel(a, b):
        return max(float((a["grads"][k].double() - b["grads"][k]).abs().max()) / max(float(b["grads"][k].abs().max()), 1e-12) for k in b["grads"] if float(b["grads"][k].abs().max()) > 1e-3)
    assert rel(r32, free64) > 10 * rel(r32, same64), (rel(r32, free64), rel(r32, same64))


def test_chained_train_steps_follow_the_reference_loss_curve():
    """golden_curve.npz = the reference's own loop (train.py:94-131) for 12 iterations of fine-tuning the COCO checkpoint,
    warm-up included; the oracle's train_step chained on its own state follows it within the loop's measured sensitivity to a
    one-ulp perturbation of the starting weights (`spread`; make_golden.py curve)."""
    g = np.load(os.path.join(GOLDEN, "golden_curve.npz"))
    c = make_golden.CURVE
    w, batches = make_golden.curve_inputs()
    anchors = [float(a) for a in np.load(os.path.join(GOLDEN, "cfg_coco.npz"))["anchors"]]
    ow, buf = w, None
    for i in range(c["iterations"]):
        x, t = batches[i % len(batches)]
        lr = oracle.warmup_lr(c["lr"], i, len(batches))
        assert lr == float(g["lr"][i])
        r = oracle.train_step(ow, torch.from_numpy(x), torch.from_numpy(t), anchors, c["classes"], lr, momentum_buf=buf)
        ow, buf = r["new_w"], r["momentum_buf"]
        for k in range(4):
            ref = float(g["curve"][i, k])
            assert abs(r["losses"][k] - ref) <= 2e-6 * abs(ref) + 8 * float(g["spread"][i, k]), (i, k, r["losses"][k], ref)
    assert float(g["curve"][-1, 3]) < 0.7 * float(g["curve"][0, 3])          # the curve does move: 14.4 -> 9.5
    for key in g.files:
        if key.startswith("final:"):
            k = key.split(":", 1)[1]
            d = np.abs(ow[k].numpy() - g[key]).max()
            assert d <= 2e-3 * max(1e-3, np.abs(g[key]).max()), (k, d)
