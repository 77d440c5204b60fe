"""SURVEY.md 8(f) row 3 on the device: ONE iteration of the reference's training loop (train.py:96-123) through the drop-in
surface - Detector in train() mode (yfv2_train_forward), compute_loss (yfv2_loss), total_loss.backward()
(yfv2_train_backward), SGD(momentum 0.949, weight_decay 0.0005).step() (yfv2_sgd_step) - on the seeded weights, images and
labels of tests/golden/golden_train.npz (what the reference's OWN modules produced: tests/golden/make_golden.py train).

What can be compared, and how.
  * Loss values are well-conditioned: held to the golden directly (1e-5).
  * Train-mode logits and gradients are not.  Batch-statistics BatchNorm over three images amplifies rounding (the
    reference's own fp32 logits are 1e-4 .. 2e-4 away from a float64 evaluation of the same step), and the gradient is a
    DISCONTINUOUS function of the ReLU decisions: a pre-activation within that rounding of zero is passed by one fp32
    execution and blocked by another, and every gradient below it moves (the reference's own fp32 run flips 8 of the
    ~1e7 decisions of case 0 against float64 and its conv1x1_3 / cls_head_3 gradients move by 1e-2 .. 8e-2 of their
    scale).  Only an implementation executing ATen's very kernels in ATen's order reproduces those digits (the CPU
    oracle does: make_golden pins it to 1e-5).
  So: (1) the device's ReLU decisions are read back (yfv2_debug_train_relu_output) and may differ from the float64
  run's only where |pre-activation| < 1e-3; (2) the float64 oracle and the fp32 oracle (= the reference's arithmetic)
  are replayed ON THE DEVICE'S DECISIONS, and the device must be AS CLOSE TO FLOAT64 AS THE REFERENCE'S ARITHMETIC IS,
  within a factor of 3, for the logits, every one of the 225 gradients, and the updated weights."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import yfv2_oracle as oracle

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden  # noqa: E402  (seeded inputs and the case list only)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden_train():
    return np.load(os.path.join(GOLDEN, "golden_train.npz"))


@pytest.mark.parametrize("ci", range(len(make_golden.TRAIN_CASES)))
def test_one_training_iteration_matches_the_reference(golden_train, ci, record_parity):
    import yolo_fastestv2_amd as yfv2
    g = golden_train
    dev = torch.device("cuda:0")
    classes, B, T, seed, lr = make_golden.TRAIN_CASES[ci]
    w, x, t = make_golden.train_case_inputs(classes, B, T, seed)
    anchors = [float(a) for a in np.load(os.path.join(GOLDEN, "cfg_coco.npz"))["anchors"]]
    cfg = {"anchor_num": 3, "classes": classes, "width": 352, "height": 352, "anchors": anchors}
    model = yfv2.Detector(classes, 3, True).to(dev)                       # train.py:70
    model.load_state_dict({k: v.clone() for k, v in w.items()})
    model.train()                                                         # train.py:96
    opt = yfv2.SGD(params=model.parameters(), lr=lr, momentum=0.949, weight_decay=0.0005)   # train.py:81-85
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[150, 250], gamma=0.1)     # train.py:88-90 accepts it
    preds = model(torch.from_numpy(x).to(dev))                            # train.py:105
    losses = yfv2.compute_loss(preds, torch.from_numpy(t).to(dev), cfg, dev)   # train.py:107
    losses[3].backward()                                                  # train.py:110
    for a, b in zip(g["loss%d" % ci], losses):
        assert abs(float(a) - float(b.detach())) <= 1e-5 * max(1.0, abs(float(a))), (float(a), float(b.detach()))
    # (1) the ReLU decisions this execution took, against the float64 run's (CPU oracle in double precision: test infrastructure)
    w64 = {k: (v.double() if v.is_floating_point() else v) for k, v in w.items()}
    x64, tt = torch.from_numpy(x).double(), torch.from_numpy(t)
    free64 = oracle.train_step(w64, x64, tt, anchors, classes, lr)
    eng = preds[0]._yfv2_engine
    decisions, n_dec, n_flip, worst_flip = {}, 0, 0, 0.0
    for name, pre in free64["pre_relu"].items():
        d = eng.debug_train_relu_output(name).reshape(pre.shape) > 0
        differ = d != (pre > 0)
        n_dec += d.numel(); n_flip += int(differ.sum())
        if differ.any():
            worst_flip = max(worst_flip, float(pre[differ].abs().max()))
        decisions[name] = d
    assert len(decisions) == 46 and worst_flip < 1e-3, (len(decisions), n_flip, worst_flip)
    # (2) float64 and the reference's fp32 arithmetic replayed on those decisions
    r64 = oracle.train_step(w64, x64, tt, anchors, classes, lr, relu_decisions=decisions)
    r32 = oracle.train_step(w, torch.from_numpy(x), tt, anchors, classes, lr, relu_decisions=decisions)
    for pi in range(6):
        t64 = r64["preds"][pi][0].numpy()
        e_ref, e_dev = np.abs(r32["preds"][pi][0].numpy() - t64).max(), np.abs(preds[pi][0].detach().cpu().numpy() - t64).max()
        assert e_dev <= 3 * e_ref + 1e-5, (pi, e_dev, e_ref)
    grads = {k: p.grad.detach().cpu() for k, p in model.named_parameters()}
    names = [str(n) for n in g["names%d" % ci]]
    assert sorted(grads) == names                                         # every parameter of the reference module got a gradient
    G = max(float(r64["grads"][k].abs().max()) for k in names)            # absolute floor for gradients that are exactly zero in
    worst_ratio, n_floor = 0.0, 0                                         # exact arithmetic (a BatchNorm shift in front of a conv + BatchNorm)
    for k in names:
        t64, ref, dev_g = r64["grads"][k].numpy(), r32["grads"][k].numpy(), grads[k].numpy()
        e_ref, e_dev = np.abs(ref - t64).max(), np.abs(dev_g - t64).max()
        assert e_dev <= 3 * e_ref + 1e-6 * G, (k, e_dev, e_ref, float(np.abs(t64).max()))
        if e_dev > 1e-6 * G:
            worst_ratio = max(worst_ratio, e_dev / max(e_ref, 1e-30))
        else:
            n_floor += 1
    opt.step()                                                            # train.py:123
    sched.step()
    after = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    for key in g.files:
        if key.startswith("after%d:" % ci):
            k = key.split(":", 1)[1]
            ref, got = g[key], after[k].numpy()
            assert got.dtype == ref.dtype, k
            if k.endswith("num_batches_tracked"):
                assert int(got) == int(ref) == 1, k
            else:       # updated weights AND the moved running statistics (which are 0.9 old + 0.1 batch: errors ~1e-6 deep in the net)
                t64 = r64["new_w"][k].numpy()
                e_ref, e_dev = np.abs(r32["new_w"][k].numpy() - t64).max(), np.abs(got - t64).max()
                assert e_dev <= 3 * e_ref + 1e-7 * max(1.0, np.abs(t64).max()), (k, e_dev, e_ref)
                if k.endswith(("running_mean", "running_var")):    # forward-only quantities: also the golden itself, loosely
                    assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), k
    record_parity("train_step_case%d" % ci, parameters=len(names), worst_error_ratio_device_over_reference_vs_float64=round(worst_ratio, 3),
                  gradients_below_the_absolute_floor=n_floor, relu_decisions=n_dec, relu_decisions_differing_from_float64=n_flip,
                  largest_pre_activation_among_those=worst_flip, loss_total_reference=float(g["loss%d" % ci][3]), loss_total_device=float(losses[3].detach()))
    # a second iteration re-uses the momentum buffers and must keep the loss sane (nothing blew up), and eval mode afterwards
    # runs the inference kernels on the UPDATED weights
    preds = model(torch.from_numpy(x).to(dev))
    l2 = yfv2.compute_loss(preds, torch.from_numpy(t).to(dev), cfg, dev)
    opt.zero_grad()
    l2[3].backward()
    opt.step()
    assert float(l2[3]) < float(losses[3]) * 1.5
    model.eval()
    with torch.no_grad():
        ev = model(torch.from_numpy(x).to(dev))
    assert all(torch.isfinite(e).all() for e in ev)


def test_gradient_accumulation_over_subdivisions():
    """train.py:122-124 steps the optimizer every `subdivisions` batches: gradients of consecutive backward calls add up in
    .grad like autograd's do."""
    import yolo_fastestv2_amd as yfv2
    dev = torch.device("cuda:0")
    classes, B, T, seed, lr = make_golden.TRAIN_CASES[1]
    w, x, t = make_golden.train_case_inputs(classes, B, T, seed)
    anchors = [float(a) for a in np.load(os.path.join(GOLDEN, "cfg_coco.npz"))["anchors"]]
    cfg = {"anchor_num": 3, "classes": classes, "width": 352, "height": 352, "anchors": anchors}
    model = yfv2.Detector(classes, 3, True).to(dev)
    model.load_state_dict(w)
    model.train()
    xs, ts = torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev)
    yfv2.compute_loss(model(xs), ts, cfg, dev)[3].backward()
    g1 = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.load_state_dict(w)                                              # same weights and statistics again (the forward moved the running stats only)
    yfv2.compute_loss(model(xs), ts, cfg, dev)[3].backward()
    for k, p in model.named_parameters():
        assert torch.allclose(p.grad, 2 * g1[k], rtol=1e-5, atol=1e-7 * float(g1[k].abs().max() + 1e-12)), k


def test_eval_after_training_runs_the_current_weights_every_time():
    """yfv2.SGD and the train-mode forward write parameters / BatchNorm statistics through raw pointers; the
    packed inference copy of the weights must be refreshed after EVERY such write, not only at the first evaluation.  train ->
    eval -> train -> eval: both evaluations must equal a fresh Detector loaded from the state_dict of that moment, bit for bit,
    and they must differ from each other (the second round of training really moved the weights).  Also a caller temporary as
    the train-mode input (ADVICE medium: the library re-reads x in backward, the engine keeps it alive)."""
    import yolo_fastestv2_amd as yfv2
    dev = torch.device("cuda:0")
    classes, B, T, seed, lr = make_golden.TRAIN_CASES[1]
    w, x, t = make_golden.train_case_inputs(classes, B, T, seed)
    anchors = [float(a) for a in np.load(os.path.join(GOLDEN, "cfg_coco.npz"))["anchors"]]
    cfg = {"anchor_num": 3, "classes": classes, "width": 352, "height": 352, "anchors": anchors}
    model = yfv2.Detector(classes, 3, True).to(dev)
    model.load_state_dict(w)
    opt = yfv2.SGD(model.parameters(), lr=lr, momentum=0.949, weight_decay=0.0005)
    ts = torch.from_numpy(t).to(dev)

    def train_once():
        model.train()
        loss = yfv2.compute_loss(model(torch.from_numpy(x).to(dev) * 1.0), ts, cfg, dev)[3]     # the input is a temporary
        junk = [torch.empty(x.size, device=dev) for _ in range(3)]                             # invite the allocator to reuse its block
        opt.zero_grad()
        loss.backward()
        del junk
        opt.step()

    def eval_now():
        model.eval()
        with torch.no_grad():
            got = [o.clone() for o in model(torch.from_numpy(x).to(dev))]
        fresh = yfv2.Detector(classes, 3, True).to(dev)
        fresh.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
        fresh.eval()
        with torch.no_grad():
            want = fresh(torch.from_numpy(x).to(dev))
        for a, b in zip(got, want):
            assert torch.equal(a, b), "eval mode ran stale weights"
        return got

    train_once()
    first = eval_now()
    train_once()
    second = eval_now()
    assert any(not torch.equal(a, b) for a, b in zip(first, second))
    # the first-conv weight gradient with a temporary input equals the one with a tensor the caller keeps
    model.load_state_dict(w)
    model.train()
    keep = torch.from_numpy(x).to(dev)
    opt.zero_grad(); yfv2.compute_loss(model(keep), ts, cfg, dev)[3].backward()
    g_keep = model.backbone.first_conv._modules["0"].weight.grad.clone()
    model.load_state_dict(w)
    opt.zero_grad()
    loss = yfv2.compute_loss(model(torch.from_numpy(x).to(dev) * 1.0), ts, cfg, dev)[3]
    junk = [torch.full((x.size,), 7.0, device=dev) for _ in range(3)]
    loss.backward()
    g_tmp = model.backbone.first_conv._modules["0"].weight.grad
    # (the weight-gradient reduction meets in float atomics: summation order varies run to run, so not bit-equal - but a stale
    # or recycled input block - here filled with 7.0 - would move every entry by orders of magnitude)
    assert torch.allclose(g_tmp, g_keep, rtol=1e-4, atol=1e-4 * float(g_keep.abs().max()))


def test_train_bind_refuses_mis_sized_buffers():
    """yfv2_train_bind checks every descriptor's element count against the handle's configuration."""
    import yolo_fastestv2_amd as yfv2
    dev = torch.device("cuda:0")
    eng = yfv2.Engine(dev, 64, 64, 5, 3, max_batch=2)
    sd = {k: v.to(dev) for k, v in yfv2.random_state_dict(1, classes=5).items() if v.is_floating_point()}
    grads = {k: torch.zeros_like(v) for k, v in sd.items() if not k.endswith(("running_mean", "running_var"))}
    eng.train_bind(sd, grads)                                              # the right sizes bind
    bad = dict(sd); bad["output_cls_layers.weight"] = torch.zeros(4 * 72, device=dev)      # one class short
    with pytest.raises(yfv2.Yfv2Error, match="output_cls_layers.weight"):
        eng.train_bind(bad, grads)
    badg = dict(grads); badg["backbone.stage3.0.branch_main.3.weight"] = torch.zeros(48 * 9 - 1, device=dev)
    with pytest.raises(yfv2.Yfv2Error, match="gradient buffer"):
        eng.train_bind(sd, badg)


@pytest.mark.parametrize("prefix,c", make_golden.CURVES, ids=["batch8", "batch64"])
def test_training_loop_follows_the_reference_loss_curve(prefix, c, record_parity):
    """train.py:94-131 line by line through the drop-in surface for 12 iterations (fine-tuning the COCO checkpoint over a
    two-batch epoch - 8 images per batch, and 64 as train.py runs it - warm-up by batch_num, step + zero_grad every iteration, MultiStepLR per epoch) against
    tests/golden/golden_curve.npz = the same loop run with the reference's own modules.  A training loop amplifies rounding:
    the golden carries `spread`, how far the REFERENCE's curve moves when its starting weights are perturbed by one fp32 ulp;
    the device's curve must stay within 8x that envelope (+ 1e-5 relative)."""
    import math
    import yolo_fastestv2_amd as yfv2
    g = np.load(os.path.join(GOLDEN, "golden_curve.npz"))
    w, batches = make_golden.curve_inputs(c)
    dev = torch.device("cuda:0")
    anchors = [float(a) for a in np.load(os.path.join(GOLDEN, "cfg_coco.npz"))["anchors"]]
    cfg = {"anchor_num": 3, "classes": c["classes"], "width": 352, "height": 352, "anchors": anchors, "learning_rate": c["lr"],
           "subdivisions": 1, "steps": [150, 250]}
    model = yfv2.Detector(cfg["classes"], cfg["anchor_num"], True).to(dev)                   # train.py:70
    model.load_state_dict({k: v.clone() for k, v in w.items()})
    optimizer = yfv2.SGD(params=model.parameters(), lr=cfg["learning_rate"], momentum=0.949, weight_decay=0.0005)
    scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=cfg["steps"], gamma=0.1)
    train_dataloader = [(torch.from_numpy(x), torch.from_numpy(t)) for x, t in batches]   # stored as float, already / 255
    curve, batch_num = [], 0
    while batch_num < c["iterations"]:
        model.train()
        for imgs, targets in train_dataloader:
            if batch_num >= c["iterations"]:
                break
            imgs = imgs.to(dev).float()
            targets = targets.to(dev)
            preds = model(imgs)
            iou_loss, obj_loss, cls_loss, total_loss = yfv2.compute_loss(preds, targets, cfg, dev)
            total_loss.backward()
            for pg in optimizer.param_groups:
                warmup_num = 5 * len(train_dataloader)
                if batch_num <= warmup_num:
                    scale = math.pow(batch_num / warmup_num, 4)
                    pg["lr"] = cfg["learning_rate"] * scale
                lr = pg["lr"]
            assert lr == float(g[prefix + "lr"][batch_num])
            if batch_num % cfg["subdivisions"] == 0:
                optimizer.step()
                optimizer.zero_grad()
            curve.append([float(v.detach()) for v in (iou_loss, obj_loss, cls_loss, total_loss)])
            batch_num += 1
        scheduler.step()
    curve = np.asarray(curve, np.float64)
    ref, spread = g[prefix + "curve"].astype(np.float64), g[prefix + "spread"].astype(np.float64)
    err = np.abs(curve - ref)
    ratio = float((err / (1e-5 * np.abs(ref) + spread)).max())
    record_parity("train_loss_curve_batch%d" % c["B"], iterations=int(c["iterations"]), total_loss_reference=[round(float(v), 6) for v in ref[:, 3]],
                  total_loss_device=[round(float(v), 6) for v in curve[:, 3]], reference_spread_under_one_ulp_perturbation=[float(v) for v in spread[:, 3]],
                  worst_error_over_envelope=round(ratio, 3))
    assert (err <= 1e-5 * np.abs(ref) + 8 * spread).all(), (err[:, 3], spread[:, 3])
    assert curve[-1, 3] < 0.9 * curve[0, 3]                                  # the curve does move
    after = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    for key in g.files:
        if key.startswith("final:") and not prefix:
            k = key.split(":", 1)[1]
            d = np.abs(after[k] - g[key]).max()
            assert d <= 2e-3 * max(1e-3, np.abs(g[key]).max()), (k, d)


def test_data_parallel_backward_issues_one_rccl_all_reduce():
    """SURVEY.md 8(e) "Training": Detector.data_parallel() on a one-rank nccl (= RCCL) group - the gradient bucket is all-reduced
    on the device, once per backward, and the iteration's result is unchanged (tests/gpu_cases/train_dp.py; the two-rank
    arithmetic runs under gloo in tests/test_abi_and_host.py)."""
    import subprocess
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_cases", "train_dp.py")
    out = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "train_dp ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_advice_r04_rebind_after_handle_recreation_pickled_optimizer_frozen_parameters():
    """Three host-side defects ADVICE r04 found in the training glue, one scenario each:
    (1) an eval call with a LARGER batch than the training batch makes Engine.ensure_batch() re-create the native handle (and
        drop its training state); the next train-mode forward must bind again instead of failing with 'yfv2_train_bind has not
        been called';
    (2) after a step, optimizer.state_dict() must pickle and the optimizer must deep-copy (the kernel-argument table of ctypes
        pointers is not part of the param groups), and a loaded state_dict carries no foreign key;
    (3) a parameter with requires_grad == False gets no .grad and is not moved by the optimizer."""
    import copy
    import io
    import yolo_fastestv2_amd as yfv2
    dev = torch.device("cuda:0")
    classes, B, T, seed, lr = make_golden.TRAIN_CASES[1]
    w, x, t = make_golden.train_case_inputs(classes, B, T, seed)
    anchors = [float(a) for a in np.load(os.path.join(GOLDEN, "cfg_coco.npz"))["anchors"]]
    cfg = {"anchor_num": 3, "classes": classes, "width": 352, "height": 352, "anchors": anchors}
    model = yfv2.Detector(classes, 3, True).to(dev)
    model.load_state_dict(w)
    frozen = "backbone.first_conv.0.weight"
    dict(model.named_parameters())[frozen].requires_grad_(False)
    before = dict(model.named_parameters())[frozen].detach().clone()
    opt = yfv2.SGD(params=model.parameters(), lr=1e-3, momentum=0.949, weight_decay=0.0005)
    xs, ts = torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev)
    model.train()
    yfv2.compute_loss(model(xs), ts, cfg, dev)[3].backward()
    assert dict(model.named_parameters())[frozen].grad is None
    assert all(p.grad is not None for k, p in model.named_parameters() if k != frozen)
    opt.step()
    assert torch.equal(dict(model.named_parameters())[frozen].detach(), before)
    # (2)
    buf = io.BytesIO()
    torch.save(opt.state_dict(), buf)
    assert all(set(g) == {"lr", "momentum", "weight_decay", "params"} or "_yfv2_table" not in g for g in opt.state_dict()["param_groups"])
    assert not any(k.startswith("_yfv2") for g in opt.state_dict()["param_groups"] for k in g)
    opt2 = copy.deepcopy(opt)
    assert len(opt2.state_dict()["state"]) == len(opt.state_dict()["state"])
    opt.zero_grad()
    # (1) eval with a larger batch than training: the handle is re-created
    model.eval()
    h_before = model.engine_for(xs).max_batch
    big = torch.rand(max(2 * B, h_before + 1), 3, 352, 352, device=dev)
    with torch.no_grad():
        model(big)
    model.train()
    loss = yfv2.compute_loss(model(xs), ts, cfg, dev)[3]
    loss.backward()
    opt.step()
    assert torch.isfinite(loss.detach()).all()


def test_single_tensor_sgd_step_is_torch_sgd():
    """yfv2_sgd_step (one tensor; since round 6 a table of one through the multi-tensor kernel): torch.optim.SGD(momentum 0.949,
    weight_decay 5e-4) on the same numbers, first step (buffer uninitialised) and a later one, to fp32 rounding (the kernel may contract
    momentum * buf + d into one FMA)."""
    import yolo_fastestv2_amd as yfv2
    dev = torch.device("cuda:0")
    eng = yfv2.Engine(dev, 352, 352, 80, 3, max_batch=1)
    torch.manual_seed(7)
    p0, g1, g2 = torch.randn(1000), torch.randn(1000), torch.randn(1000)
    p = p0.clone().to(dev)
    buf = torch.full((1000,), float("nan"), device=dev)
    eng.sgd_step(p, g1.to(dev), buf, 0.01, 0.949, 5e-4, True)
    eng.sgd_step(p, g2.to(dev), buf, 0.01, 0.949, 5e-4, False)
    torch.cuda.synchronize()
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([ref], lr=0.01, momentum=0.949, weight_decay=5e-4)
    for g in (g1, g2):
        ref.grad = g.clone()
        opt.step()
    assert torch.allclose(p.cpu(), ref.detach(), rtol=2e-6, atol=1e-7), float((p.cpu() - ref.detach()).abs().max())
