"""Drop-in evidence: the reference's own call sequences run on this package.

CPU part (no GPU): `install()` rebinding on stand-in `model.detector` / `utils.utils` modules.
GPU part (`-m gpu`): `/root/reference/test.py:18-49` and `evaluation.py:52-64` replayed statement by statement with the
module names the reference scripts use (`model.detector.Detector`, `utils.utils.load_datafile / handel_preds /
non_max_suppression / evaluation`), a stand-in for `torchsummary.summary` (forward hooks + a batch-2 forward,
evaluation.py:58), and the `export_onnx=True` output layout (model/detector.py:33-44).  Nothing here reads
/root/reference: the checkpoint and images are the committed fixtures (tests/golden/make_golden.py).
"""
import os
import sys
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN, unpack_ragged
from oracle import yfv2_oracle as oracle


def _fake_reference_modules():
    """Stand-ins for an imported reference checkout: `import model.detector, utils.utils` (test.py:7-8)."""
    model_pkg, det = types.ModuleType("model"), types.ModuleType("model.detector")
    utils_pkg, ut = types.ModuleType("utils"), types.ModuleType("utils.utils")
    model_pkg.detector, utils_pkg.utils = det, ut
    sentinel = object()
    det.Detector = sentinel
    for name in ("handel_preds", "non_max_suppression", "get_batch_statistics", "evaluation", "load_datafile", "ap_per_class", "bbox_iou"):
        setattr(ut, name, sentinel)
    return model_pkg, utils_pkg, sentinel


def test_install_rebinds_the_hot_path_and_nothing_else():
    import yolo_fastestv2_amd as yfv2
    model_pkg, utils_pkg, sentinel = _fake_reference_modules()
    yfv2.install(model_pkg.detector, utils_pkg.utils)
    assert model_pkg.detector.Detector is yfv2.Detector
    assert utils_pkg.utils.handel_preds is yfv2.handel_preds
    assert utils_pkg.utils.non_max_suppression is yfv2.non_max_suppression
    assert utils_pkg.utils.get_batch_statistics is yfv2.get_batch_statistics
    assert utils_pkg.utils.evaluation is yfv2.evaluation
    # config parsing, AP arithmetic and the IoU helper stay the reference's own
    for untouched in ("load_datafile", "ap_per_class", "bbox_iou"):
        assert getattr(utils_pkg.utils, untouched) is sentinel
    # either argument may be omitted
    m2, u2, s2 = _fake_reference_modules()
    yfv2.install(reference_utils_module=u2.utils)
    assert m2.detector.Detector is s2 and u2.utils.handel_preds is yfv2.handel_preds


def _write_reference_inputs(tmp_path, cfg, coco_weights):
    """A `.data` file in the reference's format (data/coco.data) and the checkpoint as a `.pth` (modelzoo/*.pth)."""
    names = tmp_path / "coco.names"
    names.write_text("".join("class%d\n" % i for i in range(cfg["classes"])))
    data = tmp_path / "coco.data"
    data.write_text("[name]\nmodel_name=coco\n\n[train-configure]\nepochs=300\nsteps=150,250\nbatch_size=64\nsubdivisions=1\n"
                    "learning_rate=0.001\n\n[model-configure]\npre_weights=None\nclasses=%d\nwidth=%d\nheight=%d\nanchor_num=%d\n"
                    "anchors=%s\n\n[data-configure]\ntrain=/tmp/train.txt\nval=/tmp/val.txt\nnames=%s\n"
                    % (cfg["classes"], cfg["width"], cfg["height"], cfg["anchor_num"], ",".join(repr(a) for a in cfg["anchors"]), names))
    pth = tmp_path / "coco.pth"
    torch.save({k: v.clone() for k, v in coco_weights.items()}, pth)
    return str(data), str(pth)


def _summary_stand_in(model, input_size, device):
    """What torchsummary.summary(model, input_size) does to a model (torchsummary.py): a forward hook on every
    sub-module that is not a Sequential / ModuleList / the model itself, one forward of torch.rand(2, *input_size) on
    the device, hooks removed.  Returns the number of hooks that fired and the forward's output."""
    fired, hooks = [], []

    def register(mod):
        if not isinstance(mod, (torch.nn.Sequential, torch.nn.ModuleList)) and mod is not model:
            hooks.append(mod.register_forward_hook(lambda m, i, o: fired.append(type(m).__name__)))
    model.apply(register)
    x = torch.rand(2, *input_size).type(torch.cuda.FloatTensor if device.type == "cuda" else torch.FloatTensor)
    out = model(x)
    for h in hooks:
        h.remove()
    return len(fired), out


@pytest.mark.gpu
def test_reference_test_py_replayed_line_by_line(tmp_path, cfg, coco_weights, images_u8, golden_real, monkeypatch):
    import yolo_fastestv2_amd as yfv2
    model_pkg, utils_pkg, _ = _fake_reference_modules()
    utils_pkg.utils.load_datafile = yfv2.load_datafile       # the reader the callers need (utils/utils.py:13-65)
    yfv2.install(model_pkg.detector, utils_pkg.utils)
    model, utils = model_pkg, utils_pkg                      # the names test.py uses after `import model.detector, utils.utils`
    data_path, weights_path = _write_reference_inputs(tmp_path, cfg, coco_weights)
    rows_ref, idx_ref = unpack_ragged(golden_real, "nms_03_04")

    for k in range(images_u8.shape[0]):
        # ---- test.py:18-28
        cfg_ = utils.utils.load_datafile(data_path)
        assert os.path.exists(weights_path)
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        net = model.detector.Detector(cfg_["classes"], cfg_["anchor_num"], True).to(device)
        net.load_state_dict(torch.load(weights_path, map_location=device))
        # ---- test.py:31
        net.eval()
        # ---- test.py:34-38 (res_img = the committed, already resized frame: parity is defined on identical input tensors)
        res_img = np.ascontiguousarray(images_u8[k].transpose(1, 2, 0))          # (H, W, 3) uint8, what cv2.resize returns
        img = res_img.reshape(1, cfg_["height"], cfg_["width"], 3)
        img = torch.from_numpy(img.transpose(0, 3, 1, 2))
        img = img.to(device).float() / 255.0
        # ---- test.py:42 (no torch.no_grad() there either)
        preds = net(img)
        # ---- test.py:48-49
        output = utils.utils.handel_preds(preds, cfg_, device)
        output_boxes = utils.utils.non_max_suppression(output, conf_thres=0.3, iou_thres=0.4)
        # ---- what test.py:58-69 reads: a CPU (n, 6) tensor per image, box.tolist() -> x1, y1, x2, y2, score, class
        assert isinstance(output, torch.Tensor) and output.device.type == "cpu" and tuple(output.shape) == (1, 1815, 85)
        assert len(output_boxes) == 1 and output_boxes[0].device.type == "cpu" and output_boxes[0].shape[1] == 6
        got = output_boxes[0].numpy()
        assert got.shape == rows_ref[k].shape, "image %d: %d detections, the reference has %d" % (k, got.shape[0], rows_ref[k].shape[0])
        assert np.array_equal(got[:, 5], rows_ref[k][:, 5]), "image %d: classes differ" % k
        assert np.abs(got[:, :4] - rows_ref[k][:, :4]).max(initial=0.0) <= 1e-4 * max(1.0, float(np.abs(rows_ref[k][:, :4]).max(initial=1.0)))
        assert np.abs(got[:, 4] - rows_ref[k][:, 4]).max(initial=0.0) <= 1e-5
        for box in output_boxes[0]:
            box = box.tolist()
            assert len(box) == 6 and 0 <= int(box[5]) < cfg_["classes"]
        # survivor identity in the 1815-row decode order
        _, idx = yfv2.nms_with_indices(output, 0.3, 0.4)
        assert list(np.asarray(idx[0])) == list(idx_ref[k])


@pytest.mark.gpu
def test_reference_evaluation_py_replayed(tmp_path, cfg, coco_weights, images_u8):
    """evaluation.py:52-64: Detector -> load_state_dict -> eval -> summary(model, (3,H,W)) -> utils.utils.evaluation twice
    (conf 0.01 and 0.3).  The data loader is a list of (uint8 NCHW batch, targets) pairs shaped like collate_fn's output."""
    import yolo_fastestv2_amd as yfv2
    model_pkg, utils_pkg, _ = _fake_reference_modules()
    utils_pkg.utils.load_datafile = yfv2.load_datafile
    yfv2.install(model_pkg.detector, utils_pkg.utils)
    model, utils = model_pkg, utils_pkg
    data_path, weights_path = _write_reference_inputs(tmp_path, cfg, coco_weights)
    cfg_ = utils.utils.load_datafile(data_path)
    device = torch.device("cuda")
    net = model.detector.Detector(cfg_["classes"], cfg_["anchor_num"], True).to(device)
    net.load_state_dict(torch.load(weights_path, map_location=device))
    net.eval()
    fired, out = _summary_stand_in(net, (3, cfg_["height"], cfg_["width"]), device)       # evaluation.py:58
    assert len(out) == 6 and tuple(out[0].shape) == (2, 12, 22, 22) and tuple(out[5].shape) == (2, 80, 11, 11)
    # targets: the reference's own detections at 0.3 as ground truth (class, normalised cx cy w h), one bogus object per image
    imgs = torch.from_numpy(images_u8)
    _, _, (rows, _) = oracle.detect(coco_weights, imgs.float() / 255.0, cfg["anchors"], cfg["height"], 0.3, 0.4)
    t = []
    for b, r in enumerate(rows):
        for d in r:
            t.append([b % 3, d[5], (d[0] + d[2]) / 2 / 352, (d[1] + d[3]) / 2 / 352, (d[2] - d[0]) / 352, (d[3] - d[1]) / 352])
    t = np.asarray(t, np.float32)
    loader = [(imgs[0:3], torch.from_numpy(t[:sum(len(r) for r in rows[:3])])), (imgs[3:6], torch.from_numpy(t[sum(len(r) for r in rows[:3]):]))]
    r1 = utils.utils.evaluation(loader, cfg_, net, device)                                # evaluation.py:62
    r2 = utils.utils.evaluation(loader, cfg_, net, device, 0.3)                           # evaluation.py:64
    assert len(r1) == 4 and len(r2) == 4
    _, _, AP, _ = r1
    precision, recall, _, f1 = r2
    assert AP > 0.9 and recall > 0.9 and precision > 0.9, (r1, r2)      # its own 0.3-detections are the ground truth
    print("Precision:%f Recall:%f AP:%f F1:%f" % (precision, recall, AP, f1))            # evaluation.py:65


@pytest.mark.gpu
def test_summary_stand_in_in_train_mode_runs():
    """train.py:70-71 calls summary() on a freshly built model - in TRAIN mode: hooks + a batch-2 forward.  With the training
    path in place that is a train-mode forward (batch-statistics BatchNorm): it must run and hand back the six logit maps."""
    import yolo_fastestv2_amd as yfv2
    net = yfv2.Detector(80, 3, True).to("cuda")
    assert net.training
    fired, out = _summary_stand_in(net, (3, 352, 352), torch.device("cuda"))
    assert len(out) == 6 and tuple(out[0].shape) == (2, 12, 22, 22) and tuple(out[5].shape) == (2, 80, 11, 11)
    assert all(torch.isfinite(o).all() for o in out) and out[0].requires_grad


@pytest.mark.gpu
def test_export_onnx_layout_vs_oracle(coco_weights, images_u8):
    """model/detector.py:33-44: export_onnx=True returns two NHWC maps, channels = sigmoid(12 reg) | sigmoid(3 obj) |
    softmax(classes) at 22x22 and 11x11."""
    import yolo_fastestv2_amd as yfv2
    dev = torch.device("cuda")
    net = yfv2.Detector(80, 3, True, export_onnx=True).to(dev)
    net.load_state_dict(coco_weights)
    net.eval()
    x = torch.from_numpy(images_u8[:2]).float() / 255.0
    got = net(x.to(dev))
    ref = oracle.forward(coco_weights, x)
    assert len(got) == 2
    for k, g in enumerate(got):
        r, o, c = ref[3 * k], ref[3 * k + 1], ref[3 * k + 2]
        want = torch.cat((r.sigmoid(), o.sigmoid(), torch.softmax(c, dim=1)), 1).permute(0, 2, 3, 1)
        assert tuple(g.shape) == tuple(want.shape) == (2, 22 // (k + 1), 22 // (k + 1), 95)
        assert float((g.cpu() - want).abs().max()) <= 1e-5
