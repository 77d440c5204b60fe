"""GPU parity tests: the HIP path, through the C ABI (libyfv2.so), against the
CPU oracle and the committed reference goldens.  Run with ``-m gpu`` on an MI355X.

Tolerances (BASELINE.md 5 / SURVEY.md 8(c); north_star says 1e-4 fp32):
  logits            |d| <= 1e-4
  obj / cls scores  |d| <= 1e-5
  box coords        |d| <= 1e-4 * max(1, |ref|)
  decode on IDENTICAL logits (test_decode_from_golden_logits), per field in ulps of the reference's result (DECODE_ULP):
                    centre within 2 ulp of the sigmoid it is formed from, size 12, obj 4, cls 20 (measured 1 / 6 / 2 / 10)
  logits where they are large (random-init weights): K x the REFERENCE's own fp32 error against its float64 evaluation
  NMS (identical decoded tensor in): rows and survivor indices BIT-EXACT
  end-to-end at test.py thresholds (0.3/0.4): identical survivor indices
"""
import os

import numpy as np
import pytest
import torch

import margin_nms
from conftest import GOLDEN, unpack_ragged
from oracle import yfv2_oracle as oracle

pytestmark = pytest.mark.gpu

LOGIT_KEYS = ("reg2", "obj2", "cls2", "reg3", "obj3", "cls3")
LOGIT_ATOL = 1e-4
SCORE_ATOL = 1e-5
BOX_RTOL = 1e-4


@pytest.fixture(scope="module")
def yfv2():
    import yolo_fastestv2_amd
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    assert os.path.exists(yolo_fastestv2_amd.LIB_PATH), "libyfv2.so not built"
    return yolo_fastestv2_amd


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model(yfv2, dev, coco_weights):
    m = yfv2.Detector(80, 3, True).to(dev)
    missing = m.load_state_dict(coco_weights)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.eval()


@pytest.fixture(scope="module")
def post_engine(yfv2, dev, cfg):
    eng = yfv2.get_engine(dev, cfg["height"], cfg["width"], cfg["classes"], cfg["anchor_num"])
    eng.set_anchors(cfg["anchors"])
    return eng


def _assert_decoded_close(got, ref, what=""):
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    box_ok = d[..., :4] <= BOX_RTOL * np.maximum(1.0, np.abs(ref[..., :4]))
    assert box_ok.all(), "%s box coords: worst abs %g rel %g" % (
        what, d[..., :4].max(), (d[..., :4] / np.maximum(1.0, np.abs(ref[..., :4]))).max())
    assert d[..., 4:].max() <= SCORE_ATOL, "%s scores: worst %g" % (what, d[..., 4:].max())


# ---------------------------------------------------------------------------------------
# the noise-floor rule: where an absolute tolerance does not apply - random-init weights give logits of
# magnitude 20-40 - the device is NOT granted a tolerance scaled by the logit magnitude.  It is held to a small multiple of
# the error the REFERENCE's own fp32 arithmetic (= the oracle's, pinned bit-equal) makes against a float64 evaluation of the
# same network on the same input: |device - float64| <= K * |reference_fp32 - float64|, per tensor, for the largest and for
# the rms error.  K = 2 on large samples (the bench regime: 16 images); K = 3 where a tensor has few elements and the largest
# error of either side is a noisy statistic.  FLOOR_MIN_REL: no fp32 execution is asked to be closer than two ulps of the
# tensor's magnitude on its worst element.
# ---------------------------------------------------------------------------------------
FLOOR_K = 2.0
FLOOR_K_SMALL = 3.0
FLOOR_MIN_REL = 2.0 ** -22


def _err_stats(got, exact):
    d = np.asarray(got, np.float64) - np.asarray(exact, np.float64)
    return float(np.abs(d).max()), float(np.sqrt((d * d).mean()))


def _assert_logits_within_noise_floor(got, w, x, what="", k=FLOOR_K_SMALL, err_ref=None):
    """got: six device logit maps for input x (CPU tensor) under weights w.  Returns {key: (dev max, dev rms, ref max, ref rms)}."""
    p64 = [t.numpy() for t in oracle.forward64(w, x)]
    p32 = [t.numpy() for t in oracle.forward(w, x)] if err_ref is None else None
    out = {}
    for i, key in enumerate(LOGIT_KEYS):
        g = got[i].detach().cpu().numpy()
        assert g.shape == p64[i].shape, "%s %s: shape %s vs %s" % (what, key, g.shape, p64[i].shape)
        d_max, d_rms = _err_stats(g, p64[i])
        r_max, r_rms = _err_stats(p32[i], p64[i]) if err_ref is None else (float(err_ref[key][0]), float(err_ref[key][1]))
        floor = FLOOR_MIN_REL * max(1.0, float(np.abs(p64[i]).max()))
        out[key] = (d_max, d_rms, r_max, r_rms)
        assert d_max <= k * max(r_max, floor), "%s %s: device is %.3g from float64, the reference's fp32 %.3g (bound %gx)" % (what, key, d_max, r_max, k)
        assert d_rms <= k * max(r_rms, floor / 4), "%s %s: rms %.3g from float64, the reference's fp32 %.3g (bound %gx)" % (what, key, d_rms, r_rms, k)
    return out


# ---------------------------------------------------------------------------------------
# forward
# ---------------------------------------------------------------------------------------
def test_stage_activations_vs_oracle(model, dev, images_u8, coco_weights):
    """Per-stage NHWC activations of the HIP path vs the oracle (localises a
    mismatch to stem / stage2 / stage3 / stage4 / fpn)."""
    x = torch.from_numpy(images_u8[:3]).float() / 255.0
    ref = oracle.forward_stages(coco_weights, x)
    model(x.to(dev))
    eng = model.engine_for(x.to(dev))
    for which, key in enumerate(("stem", "stage2", "c2", "c3", "s2", "s3")):
        r = ref[key].permute(0, 2, 3, 1).contiguous()  # NCHW -> NHWC
        got = eng.debug_activation(which, x.shape[0]).reshape(r.shape)
        err = float((got - r).abs().max())
        assert err <= LOGIT_ATOL, "stage %s: max abs err %g (ref max %g)" % (key, err, float(r.abs().max()))


def test_forward_real_images_vs_reference_golden(model, dev, images_u8, golden_real):
    x = (torch.from_numpy(images_u8).float() / 255.0).to(dev)
    preds = model(x)
    assert len(preds) == 6
    for p, k in zip(preds, LOGIT_KEYS):
        g = golden_real["logit_" + k]
        assert tuple(p.shape) == g.shape and p.dtype == torch.float32 and p.device.type == "cuda"
        err = np.abs(p.cpu().numpy() - g).max()
        assert err <= LOGIT_ATOL, "%s: max abs err %g" % (k, err)


def test_forward_seeded_rand_vs_reference_golden(model, dev, golden_rand):
    torch.manual_seed(1234)
    x = torch.rand(2, 3, 352, 352)
    np.testing.assert_array_equal(x.flatten()[::100003].numpy(), golden_rand["x_probe"])
    preds = model(x.to(dev))
    for p, k in zip(preds, LOGIT_KEYS):
        err = np.abs(p.cpu().numpy() - golden_rand["logit_" + k]).max()
        assert err <= LOGIT_ATOL, "%s: max abs err %g" % (k, err)


def test_forward_random_weights_odd_batches(yfv2, dev):
    """Ragged sizes: batch 1, 3 and 5 (pixel counts that are not multiples of the
    16/32/64-pixel MFMA tiles) with random-init weights, vs the oracle."""
    w = yfv2.random_state_dict(3)
    m = yfv2.Detector(80, 3, True).to(dev)
    m.load_state_dict(w)
    m.eval()
    torch.manual_seed(5)
    for B in (1, 3, 5):
        x = torch.rand(B, 3, 352, 352)
        _assert_logits_within_noise_floor(m(x.to(dev)), w, x, "B=%d" % B)


def _quirky_state_dict(yfv2, seed):
    """What fine-tuned / pruned checkpoints of the reference look like and random_state_dict never does: BatchNorm gains that went
    to exactly zero (network slimming), running variances collapsed to ~0 (a dead input: scale = gamma / sqrt(0 + 1e-5) = 316 gamma),
    whole filters zero, one filter row 1e4 times larger than its neighbours (the per-tensor power-of-two scale of the fp16x3 filter
    split is set by it: the small rows' second terms approach fp16's absolute floor), large BatchNorm shifts."""
    w = {k: v.clone() for k, v in yfv2.random_state_dict(seed).items()}
    g = torch.Generator().manual_seed(100 + seed)
    for k in sorted(w):
        if k.endswith(".running_var"):
            base = k[:-len(".running_var")]
            c = w[k].numel()
            dead = torch.rand(c, generator=g) < 0.08
            w[base + ".weight"][dead] = 0.0                                    # pruned channels: the output is the BatchNorm shift alone
            tiny = (torch.rand(c, generator=g) < 0.05) & ~dead
            w[k][tiny] = 1e-9
            w[base + ".weight"][tiny] *= 0.004                                 # ... with a gain that keeps the activation in range
            w[base + ".bias"] += 0.5 * torch.randn(c, generator=g) * (torch.rand(c, generator=g) < 0.1)
    for k in ("backbone.stage2.1.branch_main.0.weight", "backbone.stage3.4.branch_main.5.weight", "fpn.cls_head_2.block.3.weight"):
        w[k][1] = 0.0                                                          # a dead filter row
        w[k][2] *= 1e-4                                                        # a row far below the tensor's largest entry
        w[k][3, :, 0, 0] *= torch.logspace(-4, 0, w[k].shape[1])               # and four decades of range inside one row
    w["backbone.stage4.2.branch_main.3.weight"][5] = 0.0                       # a dead depthwise channel
    w["fpn.conv1x1_2.0.weight"][:, ::7] = 0.0                                  # input channels nobody reads
    return w


def test_forward_pruned_and_collapsed_batchnorm_weights(yfv2, dev):
    """Checkpoints the reference's own training produces but a seeded random init never does (zero gains, collapsed variances,
    dead filters, rows four decades below the tensor's largest entry): the device stays within the noise floor of the
    REFERENCE's fp32 arithmetic against float64 (3x: six images), and the range guard stays quiet."""
    for seed in (0, 1):
        w = _quirky_state_dict(yfv2, seed)
        m = yfv2.Detector(80, 3, True).to(dev)
        m.load_state_dict(w)
        m.eval()
        x = torch.rand(6, 3, 352, 352, generator=torch.Generator().manual_seed(40 + seed))
        got = m(x.to(dev))
        m.engine_for(x.to(dev)).check_finite("quirky weights, seed %d" % seed)
        _assert_logits_within_noise_floor(got, w, x, "quirky weights, seed %d" % seed)


def test_forward_more_images_than_compute_units(yfv2, dev):
    """Batch 300 on a 256-CU device: every one-workgroup-per-image kernel (the chains, stage3.0 / stage4.0, the towers) runs a
    second image on 44 of its workgroups, with the next image's first slice prefetched across the image boundary.  Images on
    both sides of that boundary against the oracle, and bit-identical to what the same image gives in a batch of one."""
    w = yfv2.random_state_dict(4)
    m = yfv2.Detector(80, 3, True).to(dev)
    m.load_state_dict(w)
    m.eval()
    g = torch.Generator().manual_seed(9)
    x = torch.rand(300, 3, 352, 352, generator=g)
    got = [t.cpu() for t in m(x.to(dev))]
    pick = [0, 43, 44, 255, 256, 299]
    _assert_logits_within_noise_floor([gt[pick] for gt in got], w, x[pick], "batch 300")
    for i in (44, 256, 299):
        alone = m(x[i:i + 1].to(dev))
        for gt, a, k in zip(got, alone, LOGIT_KEYS):
            assert torch.equal(gt[i], a[0].cpu()), "image %d, %s: differs from its batch-of-one result" % (i, k)


def test_batch_invariance_and_permutation(model, dev, images_u8):
    """Size-independent property at the bench batch size: every image's logits are
    bit-identical whatever its position in a 256-batch, and equal to its batch-1 result."""
    base = torch.from_numpy(images_u8).float() / 255.0
    g = torch.Generator().manual_seed(11)
    sel = torch.randint(0, base.shape[0], (256,), generator=g)
    gain = 0.8 + 0.4 * torch.rand(256, 1, 1, 1, generator=g)
    x = (base[sel] * gain).clamp(0, 1).to(dev)
    perm = torch.randperm(256, generator=g)
    a = [t.clone() for t in model(x)]
    b = model(x[perm.to(dev)])
    for ta, tb in zip(a, b):
        assert torch.equal(ta[perm.to(dev)], tb), "logits depend on batch position"
    one = model(x[17:18])
    for ta, t1 in zip(a, one):
        assert torch.equal(ta[17:18], t1), "logits depend on batch size"
    assert all(torch.isfinite(t).all() for t in a)


# ---------------------------------------------------------------------------------------
# decode
# ---------------------------------------------------------------------------------------
def _ulp_distance(got, ref):
    """distance in units of the last place between two fp32 arrays (0 where bit-equal; +0 / -0 count as equal)"""
    a = got.astype(np.float32).view(np.int32).astype(np.int64)
    b = ref.astype(np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


# Bounds = twice the largest distance ever measured on identical input logits (profiles/r06*_parity_counts.json "decode_ulp": centre
# 0.63, size 6, objectness 2, class 10), rounded up.  The box CENTRE (utils/utils.py:336-341: (2 sigmoid - 0.5 + grid) * stride) cancels near the image's
# left / top edge, so its last place says nothing there (1024 ulp for an error of 1e-7 pixels): it is held in units of the SIGMOID's
# last place, 2^-24 x 2 x stride pixels, after 1.5 ulp of the result for the two roundings behind the sigmoid.
DECODE_ULP = {"centre_in_sigmoid_ulp": 2.0, "size": 12, "obj": 4, "cls": 20}


def test_decode_from_golden_logits(post_engine, dev, cfg, golden_real, golden_rand, golden_kat, record_parity):
    """handel_preds on the SAME logits the reference decoded (goldens made by the reference's own function): the device rounds like the
    reference except for its exp / reciprocal.  Asserted in units of the last place, per column group (the
    bound used to be 1e-4 relative / 1e-5 absolute - 400 x what was documented); class probabilities below 1e-30 (the softmax of a
    logit 70 under the row's maximum) are compared absolutely - a denormal's last place says nothing."""
    worst = {}
    n0 = 3 * (cfg["height"] // 16) * (cfg["width"] // 16)
    for name, z in (("real", golden_real), ("rand", golden_rand), ("kat", golden_kat)):
        preds = [torch.from_numpy(z["logit_" + k]).to(dev) for k in LOGIT_KEYS]
        dec = post_engine.decode(preds).cpu().numpy()
        ref = z["decoded"]
        assert dec.shape == ref.shape
        _assert_decoded_close(dec, ref, name)
        d = _ulp_distance(dec, ref)
        stride = np.where(np.arange(ref.shape[1]) < n0, 16.0, 32.0)[None, :, None]
        err = np.abs(dec[..., :2].astype(np.float64) - ref[..., :2]) - 1.5 * np.spacing(np.abs(ref[..., :2]).astype(np.float32))
        centre = float((np.maximum(err, 0.0) / (2.0 ** -24 * 2.0 * stride)).max())
        tiny = np.abs(ref[..., 5:]) < 1e-30
        assert np.abs(dec[..., 5:][tiny].astype(np.float64) - ref[..., 5:][tiny]).max(initial=0.0) <= 1e-35
        worst[name] = {"centre_in_sigmoid_ulp": round(centre, 3), "size": int(d[..., 2:4].max()), "obj": int(d[..., 4].max()),
                       "cls": int(d[..., 5:][~tiny].max(initial=0)), "differing_elements": int((d > 0).sum()), "elements": int(d.size)}
    record_parity("decode_ulp", **worst)
    for name, wst in worst.items():
        assert all(wst[k] <= v for k, v in DECODE_ULP.items()), (name, wst)


def test_decode_known_answers(post_engine, dev, golden_kat):
    preds = [torch.from_numpy(golden_kat["logit_" + k]).to(dev) for k in LOGIT_KEYS]
    dec = post_engine.decode(preds).cpu().numpy()[0]
    np.testing.assert_allclose(dec[352, :4], [120.0, 88.0, 37.88, 51.48], rtol=1e-6)
    np.testing.assert_allclose(dec[1565, :4], [144.0, 112.0, 279.92, 258.87], rtol=1e-6)
    assert dec[352, 5:].argmax() == 17 and dec[1565, 5:].argmax() == 63
    np.testing.assert_allclose(dec[:, 5:].sum(1), 1.0, atol=1e-5)


def test_handel_preds_surface(yfv2, model, dev, cfg, images_u8, golden_real):
    """Reference call shape: handel_preds(preds, cfg, device) -> CPU fp32 (B,1815,85)."""
    x = (torch.from_numpy(images_u8).float() / 255.0).to(dev)
    out = yfv2.handel_preds(model(x), cfg, dev)
    assert out.device.type == "cpu" and out.dtype == torch.float32 and tuple(out.shape) == (6, 1815, 85)
    _assert_decoded_close(out.numpy(), golden_real["decoded"], "handel_preds")


def test_nms_honours_edits_of_the_tensor_handel_preds_returned(yfv2, model, dev, cfg, images_u8):
    """handel_preds leaves the device copy of its result on the CPU tensor it returns and non_max_suppression
    reuses it - but the reference's callers own that tensor: an in-place edit, through torch OR through `.numpy()` (which does not
    bump the tensor's version), must reach the NMS exactly as it would in the reference."""
    x = (torch.from_numpy(images_u8).float() / 255.0).to(dev)
    out = yfv2.handel_preds(model(x), cfg, dev)
    base = yfv2.non_max_suppression(out, 0.3, 0.4)
    assert base[0].shape[0] > 0 and base[1].shape[0] > 0
    assert all(torch.equal(a, b) for a, b in zip(base, yfv2.non_max_suppression(out, 0.3, 0.4)))      # unedited: the cached copy, same rows
    out.numpy()[0, :, 4] = 0.0                                     # numpy-side write: image 0's objectness -> nothing passes conf_thres
    edited = yfv2.non_max_suppression(out, 0.3, 0.4)
    assert edited[0].shape[0] == 0 and all(torch.equal(a, b) for a, b in zip(base[1:], edited[1:]))
    out[1, :, 4] = 0.0                                             # torch-side write
    edited = yfv2.non_max_suppression(out, 0.3, 0.4)
    assert edited[0].shape[0] == 0 and edited[1].shape[0] == 0 and all(torch.equal(a, b) for a, b in zip(base[2:], edited[2:]))
    ref_rows, _ = oracle.non_max_suppression(out.numpy(), 0.3, 0.4)          # what the reference's function returns on the edited tensor
    for b in range(len(edited)):
        assert np.array_equal(edited[b].numpy().view(np.uint32), ref_rows[b].view(np.uint32)), b


# ---------------------------------------------------------------------------------------
# NMS
# ---------------------------------------------------------------------------------------
def _check_nms_exact(yfv2, z, prefix, conf, iou):
    rows, idx = yfv2.nms_with_indices(torch.from_numpy(z["decoded"]), conf, iou)
    g_rows, g_idx = unpack_ragged(z, prefix)
    assert len(rows) == len(g_rows)
    for b in range(len(rows)):
        assert rows[b].device.type == "cpu" and rows[b].dtype == torch.float32 and rows[b].shape[1] == 6
        assert tuple(rows[b].shape) == g_rows[b].shape, (prefix, b, tuple(rows[b].shape), g_rows[b].shape)
        assert np.array_equal(rows[b].numpy().view(np.uint32), g_rows[b].view(np.uint32)), (prefix, b)
        assert np.array_equal(idx[b].numpy(), g_idx[b]), (prefix, b)


def test_nms_bit_exact_vs_reference_golden(yfv2, golden_real, golden_rand):
    for z in (golden_real, golden_rand):
        _check_nms_exact(yfv2, z, "nms_03_04", 0.3, 0.4)
        _check_nms_exact(yfv2, z, "nms_001_04", 0.01, 0.4)
        _check_nms_exact(yfv2, z, "nms_03_045", 0.3, 0.45)


def test_nms_bit_exact_stress(yfv2, golden_stress):
    """Clusters, 1815 candidates, exact ties, > max_det survivors."""
    _check_nms_exact(yfv2, golden_stress, "nms_03_04", 0.3, 0.4)
    _check_nms_exact(yfv2, golden_stress, "nms_001_04", 0.01, 0.4)
    _check_nms_exact(yfv2, golden_stress, "nms_025_06", 0.25, 0.6)


def test_nms_edge_cases(yfv2, golden_real):
    empty = torch.zeros(3, 1815, 85)
    out = yfv2.non_max_suppression(empty, 0.3, 0.4)
    assert len(out) == 3 and all(tuple(o.shape) == (0, 6) for o in out)
    # class filter (utils.py:271-272) vs the oracle
    dec = golden_real["decoded"]
    for classes in ([0], [2, 7], [79]):
        rows, idx = yfv2.nms_with_indices(torch.from_numpy(dec), 0.01, 0.4, classes=classes)
        o_rows, o_idx = oracle.non_max_suppression(dec, 0.01, 0.4, classes=classes)
        for b in range(dec.shape[0]):
            assert np.array_equal(rows[b].numpy().view(np.uint32), o_rows[b].view(np.uint32)), (classes, b)
            assert np.array_equal(idx[b].numpy(), o_idx[b])


def test_nms_idempotent_full_batch(yfv2, post_engine, dev, golden_stress):
    """Size-independent property at batch 256: NMS of the survivors alone keeps all of them,
    and the device result equals the oracle on the device's own decoded input."""
    base = torch.from_numpy(golden_stress["decoded"])
    dec = base[torch.arange(256) % base.shape[0]].clone()
    dec[:, :, 4] *= torch.linspace(0.5, 1.0, 256).view(-1, 1)
    dets, idx, cnt = post_engine.nms(dec.to(dev), 0.25, 0.45)
    rows, ids = yfv2.unpack_detections(dets, idx, cnt)
    o_rows, o_idx = oracle.non_max_suppression(dec.numpy(), 0.25, 0.45)
    for b in range(256):
        assert np.array_equal(rows[b].numpy().view(np.uint32), o_rows[b].view(np.uint32)), b
        assert np.array_equal(ids[b].numpy(), o_idx[b]), b
    keep_only = torch.zeros_like(dec)
    for b in range(256):
        keep_only[b, ids[b]] = dec[b, ids[b]]
    _, ids2 = yfv2.nms_with_indices(keep_only, 0.25, 0.45)
    for b in range(256):
        assert torch.equal(ids2[b], ids[b]), b


# ---------------------------------------------------------------------------------------
# end to end
# ---------------------------------------------------------------------------------------
def test_end_to_end_survivors_match_reference(yfv2, model, dev, cfg, images_u8, golden_real):
    """test.py flow (forward -> handel_preds -> NMS 0.3/0.4): identical survivor indices,
    rows within the box/score tolerances."""
    x = (torch.from_numpy(images_u8).float() / 255.0).to(dev)
    dec = yfv2.handel_preds(model(x), cfg, dev)
    out = yfv2.non_max_suppression(dec, conf_thres=0.3, iou_thres=0.4)
    rows, idx = yfv2.nms_with_indices(dec, 0.3, 0.4)
    g_rows, g_idx = unpack_ragged(golden_real, "nms_03_04")
    for b in range(x.shape[0]):
        assert torch.equal(out[b], rows[b])
        assert list(idx[b].numpy()) == list(g_idx[b]), "image %d: survivors %s vs reference %s" % (b, idx[b].tolist(), list(g_idx[b]))
        r, g = rows[b].numpy(), g_rows[b]
        assert (np.abs(r[:, :4] - g[:, :4]) <= BOX_RTOL * np.maximum(1, np.abs(g[:, :4]))).all()
        assert np.abs(r[:, 4] - g[:, 4]).max() <= SCORE_ATOL and np.array_equal(r[:, 5], g[:, 5])


def test_detect_fused_call_equals_three_calls(yfv2, model, dev, cfg, images_u8):
    x = (torch.from_numpy(images_u8).float() / 255.0).to(dev)
    eng = model.engine_for(x)
    eng.set_anchors(cfg["anchors"])
    d1 = eng.detect(x, 0.3, 0.4)
    d2 = eng.nms(eng.decode(eng.forward(x)), 0.3, 0.4)
    r1, i1 = yfv2.unpack_detections(*d1)
    r2, i2 = yfv2.unpack_detections(*d2)
    for a, b, c, d in zip(r1, r2, i1, i2):
        assert torch.equal(a, b) and torch.equal(c, d)


def test_errors_are_loud(yfv2, model, dev):
    with pytest.raises(RuntimeError):
        model(torch.rand(1, 3, 352, 352))  # CPU tensor: no CPU path
    with pytest.raises(ValueError):
        model.engine_for(torch.rand(1, 3, 352, 352, device=dev)).forward(torch.rand(1, 3, 320, 352, device=dev))
    model.train()
    with pytest.raises(ValueError):
        model(torch.zeros(1, 352, 352, 3, dtype=torch.uint8, device=dev))   # train mode takes train.py:101's fp32 tensor only
    model.eval()
    with pytest.raises(yfv2.Yfv2Error):
        yfv2.Engine(dev, 352, 352, 80, 3, max_batch=1).forward(torch.rand(1, 3, 352, 352, device=dev))  # no weights


def test_bench_under_torchrun_single_rank_uses_rccl(dev):
    """N>1 code path on the one GPU we have: bench.py under torch.distributed.run with one rank
    initialises the nccl (= RCCL) process group and runs the all-gather inside the timed step."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29617", os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--batch", "32", "--no-cpu-baseline", "--profile-iters", "1", "--no-extras", "--spinup-seconds", "0.3", "--blocks", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["value"] > 0 and "RCCL all-gather" in j["config"]["workload"]
    assert j["roofline"]["frac"] > 0 and j["unit"] == "images/s"
    # the line says what it ran at and what it is
    assert j["warmup"] == 1 and j["blocks"]["n"] == 3 and len(j["blocks"]["img_s"]) == 3
    assert j["blocks"]["img_s_min"] <= j["value"] <= j["blocks"]["img_s_max"]
    assert "fp16x3" in j["dtype"] and j["config"]["batches_in_flight"] == 3 and j["config"]["single_stream_img_s"] == j["single_stream_img_s"]
    for k in ("sclk_cold", "sclk_after_timed", "sclk_during_pipelined_steps", "sclk_during_single_stream_steps"):
        c = j["box"][k]
        assert 300.0 < c["sclk_mhz_min"] <= c["sclk_mhz_mean"] <= c["sclk_mhz_max"] < 3000.0, (k, c)     # MI355X: up to 2400 MHz
        assert 90.0 < c["ref_clock_mhz"] < 110.0
    assert j["box"]["sclk_after_timed"]["xcds_seen"] == 8
    assert all(r["kcycles"] > 0 for r in j["kernel_table"])


def _batch_from_reference_images(images_u8, n, seed):
    """BASELINE config 3 input: a batch built from the shipped JPEGs with deterministic
    variants (h-flip, +-16 px roll, gain in [0.8, 1.2]) so that it contains real objects."""
    base = torch.from_numpy(images_u8).float() / 255.0
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        x = base[i % base.shape[0]]
        if int(torch.randint(0, 2, (1,), generator=g)):
            x = x.flip(-1)
        dy, dx = (int(v) for v in torch.randint(-16, 17, (2,), generator=g))
        x = torch.roll(x, shifts=(dy, dx), dims=(-2, -1))
        gain = 0.8 + 0.4 * float(torch.rand(1, generator=g))
        out.append((x * gain).clamp(0, 1))
    return torch.stack(out)


def _xyxy_cls(dec_row_set):
    """(n, 85) decoded rows -> xyxy boxes (+ class * 4096 offset, utils.py:283-285), conf, class, as the reference's NMS sees them"""
    d = dec_row_set.astype(np.float32)
    p = d[:, 5:] * d[:, 4:5]
    conf, cls = p.max(1), p.argmax(1)
    half_w, half_h = d[:, 2] / np.float32(2), d[:, 3] / np.float32(2)
    box = np.stack([d[:, 0] - half_w, d[:, 1] - half_h, d[:, 0] + half_w, d[:, 1] + half_h], 1) + (cls * 4096).astype(np.float32)[:, None]
    return box, conf, cls


def _unexplained(o_dec_img, differing, thr, iou_thr):
    """Survivor differences between two valid fp32 executions are legitimate only where a decision sits on a margin
    (reference-vs-reference noise is ~1e-6 on scores): the candidate's obj / conf within 1e-4 of the
    threshold, or - for a suppression decision - an overlapping same-class candidate whose IoU with it is within 2e-3 of
    the NMS threshold, or whose conf is within 1e-5 of its own (visiting order).  Returns the rows nothing explains."""
    if not differing:
        return []
    obj = o_dec_img[:, 4]
    cand = np.flatnonzero(obj > thr - 1e-4)
    box, conf, cls = _xyxy_cls(o_dec_img[cand])
    where = {int(n): k for k, n in enumerate(cand)}
    area = (box[:, 2] - box[:, 0]) * (box[:, 3] - box[:, 1])
    bad = []
    for n in differing:
        k = where.get(int(n))
        if k is None:                                   # objectness clearly below the threshold: nobody may keep this row
            bad.append(int(n))
            continue
        if abs(float(obj[n]) - thr) < 1e-4 or abs(float(conf[k]) - thr) < 1e-4:
            continue
        iw = np.clip(np.minimum(box[:, 2], box[k, 2]) - np.maximum(box[:, 0], box[k, 0]), 0, None)
        ih = np.clip(np.minimum(box[:, 3], box[k, 3]) - np.maximum(box[:, 1], box[k, 1]), 0, None)
        iou = iw * ih / (area + area[k] - iw * ih + 1e-30)
        near_iou = np.abs(iou - iou_thr) < 2e-3
        tie = (np.abs(conf - conf[k]) < 1e-5) & (iou > iou_thr - 2e-3)
        near_iou[k] = tie[k] = False
        # a neighbour that is itself on a margin can flip this one (chains of suppression): look one step out
        flip = near_iou | tie
        if not flip.any():
            second = iou > iou_thr - 2e-3
            second[k] = False
            for m in np.flatnonzero(second):
                iw2 = np.clip(np.minimum(box[:, 2], box[m, 2]) - np.maximum(box[:, 0], box[m, 0]), 0, None)
                ih2 = np.clip(np.minimum(box[:, 3], box[m, 3]) - np.maximum(box[:, 1], box[m, 1]), 0, None)
                iou2 = iw2 * ih2 / (area + area[m] - iw2 * ih2 + 1e-30)
                t2 = (np.abs(iou2 - iou_thr) < 2e-3) | ((np.abs(conf - conf[m]) < 1e-5) & (iou2 > iou_thr - 2e-3))
                t2[m] = False
                if t2.any() or abs(float(conf[m]) - thr) < 1e-4:
                    flip[m] = True
                    break
        if not flip.any():
            bad.append(int(n))
    return bad


@pytest.mark.parametrize("conf_thres", [0.3, 0.01], ids=["test.py-0.3", "evaluation-0.01"])
def test_end_to_end_batch256_detection_set_vs_oracle(yfv2, model, dev, cfg, images_u8, coco_weights, conf_thres, record_parity):
    """BASELINE config 3: batch 256, COCO weights, forward + decode + NMS at the test.py thresholds (0.3 / 0.4) and at
    evaluation()'s (0.01 / 0.4).  Without COCO val the mAP check is detection-set parity: the GPU survivors must equal the
    CPU oracle's.  A difference is accepted ONLY where `_unexplained` finds the decision on a numerical margin; the count of
    unexplained differences must be zero (printed with -s), and the rows of common survivors must agree to tolerance."""
    x = _batch_from_reference_images(images_u8, 256, seed=3)
    eng = model.engine_for(x.to(dev))
    eng.set_anchors(cfg["anchors"])
    rows, idx = yfv2.unpack_detections(*eng.detect(x.to(dev), conf_thres, 0.4))
    _, o_dec, (o_rows, o_idx) = oracle.detect(coco_weights, x, cfg["anchors"], cfg["height"], conf_thres, 0.4)
    n_det = sum(len(i) for i in o_idx)
    assert n_det > 500, "the synthetic batch must contain real detections (got %d)" % n_det
    n_diff, bad, n_uncertain, interval_bad = 0, [], 0, []
    for b in range(256):
        got, ref = set(idx[b].tolist()), set(int(v) for v in o_idx[b])
        n_diff += len(got ^ ref)
        bad += [(b, n) for n in _unexplained(o_dec[b], sorted(got ^ ref), conf_thres, 0.4)]
        # the cascade-aware interval rule (tests/margin_nms.py): what the device MUST and MAY report for this image
        m = margin_nms.check(o_dec[b], idx[b].tolist(), conf_thres, 0.4)
        n_uncertain += m["n_uncertain"]
        interval_bad += [(b, n) for n in m["missing"] + m["forbidden"]]
        common = sorted(got & ref)
        if common:
            gi = {int(n): k for k, n in enumerate(idx[b].tolist())}
            ri = {int(n): k for k, n in enumerate(o_idx[b])}
            g = rows[b].numpy()[[gi[n] for n in common]]
            r = o_rows[b][[ri[n] for n in common]]
            assert (np.abs(g[:, :4] - r[:, :4]) <= BOX_RTOL * np.maximum(1, np.abs(r[:, :4]))).all()
            assert np.abs(g[:, 4] - r[:, 4]).max() <= SCORE_ATOL and np.array_equal(g[:, 5], r[:, 5])
    print("conf %.2f: %d oracle detections, %d survivor differences, %d unexplained" % (conf_thres, n_det, n_diff, len(bad)))
    record_parity("coco_batch256_conf%.2f_iou0.40" % conf_thres, n_det=n_det, n_diff=n_diff, explained=n_diff - len(bad),
                  unexplained=len(bad), interval_rule_violations=len(interval_bad), rows_on_a_margin=n_uncertain)
    assert not bad, "%d unexplained survivor differences out of %d detections (%d on numerical margins): %s" % (
        len(bad), n_det, n_diff - len(bad), bad[:8])
    assert not interval_bad, "%d rows violate the interval rule of tests/margin_nms.py: %s" % (len(interval_bad), interval_bad[:8])
    # the differences are bounded too, not only explained: 0.3 - the margins are wide; 0.01 - r03a measured
    # 5 differences out of 10 984 detections, 517 rows on a margin (profiles/r03a_parity_counts.json); the bound leaves a
    # factor of slack for box-to-box differences in the last bit of a score
    assert n_diff <= (2 if conf_thres == 0.3 else N_DIFF_BOUND_001), "%d survivor differences at conf %.2f" % (n_diff, conf_thres)


# 2 x the worst ever measured: 5 differences out of 10 984 detections in every evidence run of rounds 3-5 (thirteen runs on
# different boxes and builds, profiles/*_parity_counts.json: the device is deterministic, the count did not move once); all five
# sit on a numerical margin.  (Rounds 3-4 allowed 24.)
N_DIFF_BOUND_001 = 10


def test_fused_post_pinned_in_the_bench_regime(yfv2, dev, record_parity):
    """the launch inside bench.py's `value` is nms_kernel<2> (decode + NMS fused) on the bench's own
    workload - seeded random-init weights, torch.rand images generated on the device with seed 1000, conf 0.3 / IoU 0.4,
    B = 256, where EVERY image reports the maximum of 300 detections.  (1) the fused launch equals the three-call path
    (yfv2_forward -> yfv2_decode -> yfv2_nms, whose NMS is pinned bit-exact against the reference goldens) bit for bit - rows,
    indices, counts - on all 256 images; (2) the three-call NMS equals the oracle's NMS on the device's own decoded tensor
    (bit-exact, 32 images); (3) end to end against the CPU oracle's forward + decode + NMS on 16 images under the interval
    rule of tests/margin_nms.py (a decision may differ only where it sits on a numerical margin), counts recorded."""
    import bench
    sd = yfv2.random_state_dict(0)
    eng = yfv2.Engine(dev, 352, 352, 80, 3, anchors=bench.ANCHORS, max_batch=256)
    eng.load_state_dict(sd)
    g = torch.Generator(device=dev).manual_seed(1000)
    x = torch.rand(256, 3, 352, 352, device=dev, generator=g)
    d1, i1, c1 = eng.detect(x, 0.3, 0.4)
    dec = eng.decode(eng.forward(x))
    d2, i2, c2 = eng.nms(dec, 0.3, 0.4)
    assert torch.equal(c1, c2), "fused and three-call counts differ"
    assert int(c1.min()) == 300 and int(c1.max()) == 300, "the bench workload is the 300-detections worst case (got %d..%d)" % (int(c1.min()), int(c1.max()))
    assert torch.equal(i1, i2), "fused and three-call survivor indices differ"
    assert torch.equal(d1.view(torch.int32), d2.view(torch.int32)), "fused and three-call rows differ"
    dec_h = dec[:32].cpu().numpy()
    o_rows, o_idx = oracle.non_max_suppression(dec_h, 0.3, 0.4)
    rows, ids = yfv2.unpack_detections(d1[:32], i1[:32], c1[:32])
    for b in range(32):
        assert np.array_equal(rows[b].numpy().view(np.uint32), o_rows[b].view(np.uint32)), b
        assert np.array_equal(ids[b].numpy(), o_idx[b]), b
    n16 = 16
    # (3a) identical logits in: the oracle's decode + NMS of the device's OWN logits against the fused launch - the margins of
    # the interval rule only have to absorb the decode's <= 2 ulp: the defaults are generous
    logits = [t[:n16].cpu() for t in eng.forward(x)]
    own_dec = oracle.decode(logits, bench.ANCHORS, 352)
    _assert_decoded_close(dec_h[:n16], own_dec, "bench regime decode (identical logits)")
    _, own_idx = oracle.non_max_suppression(own_dec, 0.3, 0.4)
    own_diff, own_viol, own_unc = 0, [], 0
    for b in range(n16):
        own_diff += len(set(ids[b].tolist()) ^ set(int(v) for v in own_idx[b]))
        m = margin_nms.check(own_dec[b], ids[b].tolist(), 0.3, 0.4)
        own_unc += m["n_uncertain"]
        own_viol += [(b, n) for n in m["missing"] + m["forbidden"]]
    # (3b) end to end against the CPU oracle's forward + decode + NMS, under the noise-floor rule: the device's logits and its
    # decoded tensor are held to FLOOR_K x the error of the reference's own fp32 arithmetic against float64 on these very
    # images, and the margins of the interval rule follow from that floor - NOT from the device's own distance to the oracle
    xh = x[:n16].cpu()
    fl = _bench_regime_floor_check(sd, xh, logits, dec_h[:n16], [ids[b].tolist() for b in range(n16)], err_ref=None)
    record_parity("bench_regime_random_weights_conf0.30_iou0.40", images_fused_vs_three_call=256, images_nms_vs_oracle_bitexact=32,
                  images_vs_oracle=n16, identical_logits_n_diff=own_diff, identical_logits_rows_on_a_margin=own_unc,
                  identical_logits_interval_rule_violations=len(own_viol), **fl["record"])
    assert not own_viol, "identical logits: %d rows violate the interval rule: %s" % (len(own_viol), own_viol[:8])
    assert not fl["violations"], "%d rows violate the interval rule: %s" % (len(fl["violations"]), fl["violations"][:8])


def _bench_regime_floor_check(sd, xh, dev_logits, dev_dec, dev_ids, err_ref):
    """The noise-floor rule end to end on `xh` (CPU images) under weights `sd`: device logits (six CPU tensors), device decoded
    rows (numpy) and device survivor ids per image.  err_ref: the REFERENCE's own error statistics against float64 from
    golden_floor.npz ({key: (max, rms)}, "decoded": (box max, box rms, score max, score rms)) or None = measure them here with
    the oracle (= the reference's arithmetic, pinned bit-equal by make_golden.py).  Asserts the logits and the decoded tensor;
    returns the record fields and the interval-rule violations (asserted by the caller after recording)."""
    import bench
    n = xh.shape[0]
    p64 = oracle.forward64(sd, xh)
    dec64 = oracle.decode64(p64, bench.ANCHORS, 352)
    p32 = oracle.forward(sd, xh)
    o_dec = oracle.decode(p32, bench.ANCHORS, 352)
    _, e_idx = oracle.non_max_suppression(o_dec, 0.3, 0.4)
    st = _assert_logits_within_noise_floor(dev_logits, sd, xh, "bench regime", k=FLOOR_K, err_ref=err_ref)

    def dstats(got):
        d = np.asarray(got, np.float64) - dec64
        db = d[..., :4] / np.maximum(1.0, np.abs(dec64[..., :4]))
        return (float(np.abs(db).max()), float(np.sqrt((db * db).mean())), float(np.abs(d[..., 4:]).max()), float(np.sqrt((d[..., 4:] ** 2).mean())))
    dd = dstats(dev_dec)
    rd = dstats(o_dec) if err_ref is None else tuple(float(v) for v in err_ref["decoded"])
    for name, a, b in zip(("box max", "box rms", "score max", "score rms"), dd, rd):
        assert a <= FLOOR_K * b, "bench regime decoded %s: device %.3g from float64, the reference's fp32 %.3g (bound %gx)" % (name, a, b, FLOOR_K)
    # margins of the interval rule around the ORACLE's decoded rows: the device sits within FLOOR_K floors of float64 and the
    # oracle within one, so two rows' scores differ by at most (FLOOR_K + 1) floors; conf = obj * cls moves by at most twice
    # a score's move; a tie is open when two confs can cross
    e_s, e_b = (FLOOR_K + 1) * rd[2], (FLOOR_K + 1) * rd[0]
    eps = dict(eps_conf=2 * e_s, eps_tie=4 * e_s, eps_iou=margin_nms.EPS_IOU, box_rtol=e_b)
    n_diff = n_uncertain = n_cand = 0
    violations = []
    for b in range(n):
        got, ref = set(dev_ids[b]), set(int(v) for v in e_idx[b])
        n_diff += len(got ^ ref)
        m = margin_nms.check(o_dec[b], dev_ids[b], 0.3, 0.4, **eps)
        n_uncertain += m["n_uncertain"]; n_cand += m["n_candidates"]
        violations += [(b, r) for r in m["missing"] + m["forbidden"]]
    worst = max(st[k][0] / max(st[k][2], 1e-30) for k in LOGIT_KEYS)
    rec = dict(end_to_end_n_det=sum(len(i) for i in e_idx), end_to_end_candidates=n_cand, end_to_end_n_diff=n_diff,
               end_to_end_rows_on_a_margin=n_uncertain, end_to_end_interval_rule_violations=len(violations),
               logit_scale=round(max(float(t.abs().max()) for t in p64), 2),
               logit_err_vs_float64_device_max=max(st[k][0] for k in LOGIT_KEYS), logit_err_vs_float64_reference_max=max(st[k][2] for k in LOGIT_KEYS),
               logit_err_vs_float64_device_rms=max(st[k][1] for k in LOGIT_KEYS), logit_err_vs_float64_reference_rms=max(st[k][3] for k in LOGIT_KEYS),
               worst_logit_ratio_device_over_reference=round(worst, 3), floor_k=FLOOR_K,
               box_rel_err_vs_float64_device=dd[0], box_rel_err_vs_float64_reference=rd[0],
               score_err_vs_float64_device=dd[2], score_err_vs_float64_reference=rd[2],
               margins=dict(eps_conf=eps["eps_conf"], eps_tie=eps["eps_tie"], box_rtol=eps["box_rtol"]))
    return {"record": rec, "violations": violations}


def test_bench_regime_within_the_reference_noise_floor(yfv2, dev, golden_floor, record_parity):
    """VERDICT r03 weak 1.  The bench's regime (random_state_dict(0), U[0,1) images, conf 0.3 / IoU 0.4, 300 detections per
    image) on the workload tests/golden/golden_floor.npz was made on, where the REFERENCE's own modules were run in fp32 and
    in float64 (make_golden.py floor): (1) this host's float64 oracle reproduces the reference's float64 logits (pins the
    yardstick); (2) the device's six logit maps and its decoded tensor are within FLOOR_K x of the reference's recorded fp32
    error against float64 - largest and rms error, per tensor; (3) the survivors satisfy the interval rule with margins derived
    from that recorded floor."""
    import bench
    from conftest import floor_inputs
    sd, xh = floor_inputs(golden_floor)
    keep = int(golden_floor["keep"])
    for k, t in zip(LOGIT_KEYS, oracle.forward64(sd, xh[:keep])):
        assert np.abs(t.numpy() - golden_floor["logit64_" + k]).max() <= 1e-9, "float64 oracle on this host != the reference's float64 run: %s" % k
    eng = yfv2.Engine(dev, 352, 352, 80, 3, anchors=bench.ANCHORS, max_batch=xh.shape[0])
    eng.load_state_dict(sd)
    x = xh.to(dev)
    logits = [t.cpu() for t in eng.forward(x)]
    dec = eng.decode(eng.forward(x)).cpu().numpy()
    _, ids = yfv2.unpack_detections(*eng.detect(x, 0.3, 0.4))
    err_ref = {k: golden_floor["err_" + k] for k in LOGIT_KEYS}
    err_ref["decoded"] = golden_floor["err_decoded"]
    fl = _bench_regime_floor_check(sd, xh, logits, dec, [i.tolist() for i in ids], err_ref)
    record_parity("bench_regime_noise_floor_golden", images=int(xh.shape[0]), **fl["record"])
    assert not fl["violations"], "%d rows violate the interval rule: %s" % (len(fl["violations"]), fl["violations"][:8])


def test_other_input_size_320(yfv2, dev):
    """A non-default size: 320x320 (40/20/10 maps, 1500 decode rows) through the same fused kernels
    (or their layer-by-layer fallback where a static bound does not fit), vs the oracle."""
    w = yfv2.random_state_dict(11)
    m = yfv2.Detector(80, 3, True).to(dev)
    m.load_state_dict(w)
    m.eval()
    torch.manual_seed(2)
    x = torch.rand(3, 3, 320, 320)
    got = m(x.to(dev))
    _assert_logits_within_noise_floor(got, w, x, "320x320")
    anchors = [12.64, 19.39, 37.88, 51.48, 55.71, 138.31, 126.91, 78.23, 131.57, 214.55, 279.92, 258.87]
    cfg320 = {"height": 320, "width": 320, "anchor_num": 3, "anchors": anchors}
    dec = yfv2.handel_preds(got, cfg320, dev)
    assert tuple(dec.shape) == (3, 1500, 85)
    o_dec = oracle.decode([t.cpu() for t in got], anchors, 320)
    _assert_decoded_close(dec.numpy(), o_dec, "320x320")
    rows, idx = yfv2.nms_with_indices(dec, 0.3, 0.4)
    o_rows, o_idx = oracle.non_max_suppression(dec.numpy(), 0.3, 0.4)
    for b in range(3):
        assert np.array_equal(rows[b].numpy().view(np.uint32), o_rows[b].view(np.uint32)) and np.array_equal(idx[b].numpy(), o_idx[b])


@pytest.mark.parametrize("env", [{"YFV2_FUSED": "0"}, {"YFV2_BF6": "0"}, {"YFV2_POSTFUSE": "0"}, {"YFV2_TPAIR": "0"}, {"YFV2_FRONT": "0"}, {"YFV2_FUSED": "0", "YFV2_BF6": "0"}],
                         ids=["layer-by-layer", "fp32-mfma-everywhere", "two-launch-post", "tower-halves-as-four-launches", "stem-and-stage2.0-as-two-launches",
                              "layer-by-layer-on-the-fp32-mfma"])
def test_fallback_plans_match_oracle(yfv2, dev, images_u8, coco_weights, env):
    """The three plan switches that remain: everything layer by layer - also what a shape outside a fused
    kernel's static bounds gets, block by block; every pointwise conv on the fp32 MFMA; decode and NMS as two launches.
    Same logits as the oracle; the two-launch post must report exactly what the three separate calls report."""
    x = (torch.from_numpy(images_u8[:3]).float() / 255.0)
    ref = oracle.forward(coco_weights, x)
    sd = {k: torch.as_tensor(np.asarray(v)) for k, v in coco_weights.items()}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        eng = yfv2.Engine(dev, 352, 352, 80, 3, max_batch=4)   # the plan is built from the environment at load time
        eng.load_state_dict(sd)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    names = [s["name"] for s in eng.stages()]
    if "YFV2_FUSED" in env and "YFV2_BF6" in env:
        assert len(names) >= 70 and not any("chain of" in n or "lane-per-pixel" in n for n in names), names
    elif "YFV2_FUSED" in env:
        assert len(names) >= 70 and not any("chain of" in n or "lane-per-pixel" in n for n in names), names
    elif "YFV2_BF6" in env:
        assert not any("chain of 7" in n for n in names) and any("resident in LDS" in n for n in names), names
    elif "YFV2_TPAIR" in env:
        assert len(names) == 15 and not any("side by side" in n for n in names), names   # cls a, cls b, reg a, reg b at 22x22
    elif "YFV2_FRONT" in env:
        assert len(names) == 13 and sum("four halves of an image" in n for n in names) == 1 and names[0].startswith("stem conv"), names
    else:
        assert len(names) == 12 and sum("four halves of an image" in n for n in names) == 1 and names[0].startswith("stem + backbone.stage2.0 in one launch"), names
        eng.set_anchors([12.64, 19.39, 37.88, 51.48, 55.71, 138.31, 126.91, 78.23, 131.57, 214.55, 279.92, 258.87])
        r1, i1 = yfv2.unpack_detections(*eng.detect(x.to(dev), 0.3, 0.4))                 # decode_kernel<compact> + nms_kernel<1>
        r2, i2 = yfv2.unpack_detections(*eng.nms(eng.decode(eng.forward(x.to(dev))), 0.3, 0.4))
        for a, b, c, d in zip(r1, r2, i1, i2):
            assert torch.equal(a, b) and torch.equal(c, d) and len(a) > 0
    got = eng.forward(x.to(dev))
    for g, r, k in zip(got, ref, LOGIT_KEYS):
        err = float((g.cpu() - r).abs().max())
        assert err <= LOGIT_ATOL, "%s: max abs err %g" % (k, err)
    # the uint8 (B,H,W,3) entry point of the same plan (the fp32-matrix plans run their own uint8 stem, stem_px_kernel<.., U8>)
    x8 = torch.from_numpy(images_u8[:3]).permute(0, 2, 3, 1).contiguous()
    got8 = eng.forward(x8.to(dev))
    for g, r, k in zip(got8, ref, LOGIT_KEYS):
        err = float((g.cpu() - r).abs().max())
        assert err <= LOGIT_ATOL, "uint8 input, %s: max abs err %g" % (k, err)


def test_narrow_class_head_level_as_one_launch(yfv2, dev):
    """towerp_kernel<1, true>: a class head of at most 15 classes (obj + cls = one 16-channel tile) on a batch that fills the chip - the
    whole 22x22 level as ONE launch in its one-tile instantiation.  The same images in a batch of 8 go through the two-launch form
    (towerp_kernel<0 / 1, false>, pinned against the oracle by the class-count cases): the logits must be bit-identical."""
    w = yfv2.random_state_dict(21, classes=5)
    m = yfv2.Detector(5, 3, True).to(dev)
    m.load_state_dict(w)
    m.eval()
    torch.manual_seed(4)
    x = torch.rand(136, 3, 352, 352)
    big = [t.cpu() for t in m(x.to(dev))]
    names = [s["name"] for s in m.engine_for(x.to(dev)).stages()]
    assert sum("four halves of an image" in n for n in names) == 1, names
    small = [t.cpu() for t in m(x[:8].to(dev))]
    for a, b, k in zip(big, small, LOGIT_KEYS):
        assert torch.equal(a[:8], b), k
    _assert_logits_within_noise_floor([t[:3] for t in big], w, x[:3], "5 classes, batch 136")


def test_uint8_hwc_input_matches_float_path(yfv2, model, dev, images_u8, coco_weights, cfg):
    """the pre-process of test.py:34-38 (HWC uint8 -> NCHW fp32 / 255) inside the stem kernel.
    images_u8 is stored NCHW; the entry point takes the decoder's HWC layout.  Same logits as the oracle on the
    float()/255 tensor (the 1/255 is folded into the filter: rounding-level differences only), same survivors as
    the fp32 entry point end to end."""
    x_chw = torch.from_numpy(images_u8[:5])
    x_hwc = x_chw.permute(0, 2, 3, 1).contiguous()
    assert x_hwc.dtype == torch.uint8 and tuple(x_hwc.shape[1:]) == (352, 352, 3)
    ref = oracle.forward(coco_weights, x_chw.float() / 255.0)
    got = model(x_hwc.to(dev))
    for g, r, k in zip(got, ref, LOGIT_KEYS):
        assert tuple(g.shape) == tuple(r.shape)
        err = float((g.cpu() - r).abs().max())
        assert err <= LOGIT_ATOL, "%s: max abs err %g" % (k, err)
    eng = model.engine_for(x_hwc.to(dev))
    eng.set_anchors(cfg["anchors"])
    d8, i8, c8 = eng.detect(x_hwc.to(dev), 0.3, 0.4)
    df, if_, cf = eng.detect((x_chw.float() / 255.0).to(dev), 0.3, 0.4)
    assert torch.equal(c8, cf)
    for b in range(x_hwc.shape[0]):
        n = int(cf[b])
        assert torch.equal(i8[b, :n], if_[b, :n])
        assert float((d8[b, :n] - df[b, :n]).abs().max()) <= 1e-3 if n else True
    with pytest.raises(ValueError):
        eng.forward(x_chw.to(dev))   # uint8 in NCHW is not a supported layout


def test_non_square_input_288x384(yfv2, dev):
    """Height != width (36/18/9 x 48/24/12 maps): strips, bands and halo masks of the lane-per-pixel kernels and the
    tile bounds of the LDS kernels are all per-axis."""
    w = yfv2.random_state_dict(5)
    m = yfv2.Detector(80, 3, True).to(dev)
    m.load_state_dict(w)
    m.eval()
    torch.manual_seed(4)
    x = torch.rand(2, 3, 288, 384)
    _assert_logits_within_noise_floor(m(x.to(dev)), w, x, "288x384")


@pytest.mark.parametrize("hw", [(32, 32), (64, 96), (352, 32), (96, 384)])
def test_small_and_strip_shaped_inputs(yfv2, dev, hw):
    """The smallest maps the configuration check admits (a 2x2 / 1x1 pair of detection maps at 32x32) and strip-shaped
    ones: one-row bands, partial pixel tiles and the fallbacks of the kernels whose static bounds these shapes miss
    (the stage-3 chain needs >= 2 rows of slots, the lane-per-pixel stage 2 a minimum width) against the oracle."""
    w = yfv2.random_state_dict(7)
    m = yfv2.Detector(80, 3, True).to(dev)
    m.load_state_dict(w)
    m.eval()
    torch.manual_seed(hw[0] + hw[1])
    x = torch.rand(3, 3, hw[0], hw[1])
    _assert_logits_within_noise_floor(m(x.to(dev)), w, x, "%dx%d" % hw)


@pytest.mark.parametrize("hw,B", [((32, 32), 3), ((64, 96), 3), ((352, 32), 2), ((96, 384), 2), ((288, 384), 2), ((320, 320), 5), ((352, 352), 9)])
def test_uint8_entry_sizes_and_strips_vs_oracle(yfv2, dev, hw, B):
    """stem_h3u_kernel (uint8 HWC pixels on the f16 matrix cores) at the sizes the fp32 stem is tested at: strips with a
    partial last lane group, one-band images, the halo lane between strips, the first / last rows' padding - logits against
    the oracle on test.py:38's float() / 255 tensor, extreme pixels (0 and 255 runs) included."""
    w = yfv2.random_state_dict(13)
    m = yfv2.Detector(80, 3, True).to(dev)
    m.load_state_dict(w)
    m.eval()
    g = torch.Generator().manual_seed(hw[0] * 7 + hw[1])
    x = torch.randint(0, 256, (B, hw[0], hw[1], 3), generator=g, dtype=torch.uint8)
    x[0, : hw[0] // 2] = 255                      # saturated / black halves: the largest accumulators, exact zeros
    x[B - 1, :, : hw[1] // 2] = 0
    # float64 of the SAME fp32 input tensor the reference would see (test.py:38: float() / 255 in fp32)
    _assert_logits_within_noise_floor(m(x.to(dev)), w, x.permute(0, 3, 1, 2).float() / 255.0, "uint8 %dx%d" % hw)


def test_batch_statistics_bit_exact_vs_reference_golden(yfv2, dev, golden_stats):
    """evaluation()'s matching loop (utils.py:194-230) as one kernel launch; flags identical to
    the ones the reference function produced on the same detections / targets (jittered copies, twins with tied
    IoU, foreign labels, an image without targets, interleaved image order), through both surfaces."""
    dets, _ = unpack_ragged(golden_stats, "dets")
    targets = torch.from_numpy(golden_stats["targets"])
    outs = [torch.from_numpy(d) for d in dets]
    for thr, key in ((0.5, "tp_050"), (0.75, "tp_075")):
        got = yfv2.get_batch_statistics(outs, targets, thr, dev)
        assert len(got) == len(outs)
        flat = np.concatenate([t for t, _, _ in got])
        assert flat.dtype == np.float64
        assert np.array_equal(flat.astype(np.uint8), golden_stats[key])
        for (t, sc, lb), o in zip(got, outs):
            assert torch.equal(sc, o[:, 4]) and torch.equal(lb, o[:, -1])
    # None entries are skipped like in the reference; no targets at all -> all zeros
    got = yfv2.get_batch_statistics([None, outs[1]], targets, 0.5, dev)
    assert len(got) == 1
    got = yfv2.get_batch_statistics(outs[:2], torch.zeros((0, 6)), 0.5, dev)
    assert all(t.sum() == 0 for t, _, _ in got)


def test_cpp_host_class_matches_python_surface(yfv2, model, dev, cfg, images_u8, coco_weights, tmp_path):
    """include/yfv2.hpp (the counterpart of the reference's ncnn sample class, no Python in the
    process) must report exactly the detections of the Python surface for the same uint8 image: same survivors, same
    scores, boxes = the float boxes scaled to the source image and truncated like yolo-fastestv2.cpp's int casts.
    Case 1: source already 352x352 (no resize, scale 1).  Case 2: a 2x nearest-upsampled source (704x704): the device
    resize maps it back onto exactly the 352x352 image, boxes come back scaled by 2.  Case 3: a 640x480 frame."""
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "yfv2_cpp_test")
    if not os.path.exists(exe):      # normally prebuilt by __graft_entry__.build() and shipped with the tree
        import __graft_entry__
        __graft_entry__.build()
    assert os.path.exists(exe), "tests/cpp/yfv2_cpp_test missing: run __graft_entry__.build()"
    wpath = str(tmp_path / "coco.yfv2w")
    assert yfv2.export_weights(coco_weights, wpath) > 0
    anchors = ",".join(repr(float(a)) for a in cfg["anchors"])
    hwc = np.ascontiguousarray(images_u8[0].transpose(1, 2, 0))

    def run(img, cols, rows):
        ipath = str(tmp_path / ("img_%dx%d.raw" % (cols, rows)))
        img.tofile(ipath)
        r = subprocess.run([exe, wpath, anchors, ipath, str(cols), str(rows), "0.3", "0.4"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        return [ln.split() for ln in r.stdout.strip().splitlines() if ln.strip()]

    x = torch.from_numpy(hwc)[None].to(dev)
    eng = model.engine_for(x)
    d, i, c = eng.detect(x, 0.3, 0.4)
    n = int(c[0])
    assert n > 0
    want = d[0, :n].cpu().numpy()

    got = run(hwc, 352, 352)
    assert len(got) == n
    for g, w in zip(got, want):
        assert [int(v) for v in g[:4]] == [int(np.float32(v)) for v in w[:4]]
        assert int(g[4]) == int(w[5])
        assert np.float32(float(g[5])) == np.float32(w[4])

    # Case 3: an arbitrary frame size (480 rows x 640 columns): same detections as Engine.resize + Engine.detect, boxes
    # scaled by 640/352, 480/352 in fp32 and truncated.
    rng = np.random.default_rng(3)
    frame = oracle.resize_linear_u8(hwc, 640, 480)                       # just a plausible 640x480 picture
    frame = np.clip(frame.astype(int) + rng.integers(-3, 4, frame.shape), 0, 255).astype(np.uint8)
    f_dev = torch.from_numpy(frame)[None].to(dev)
    d3, _, c3 = eng.detect(eng.resize(f_dev), 0.3, 0.4)
    n3 = int(c3[0])
    got3 = run(frame, 640, 480)
    assert len(got3) == n3 and n3 > 0
    sw, sh = np.float32(640) / np.float32(352), np.float32(480) / np.float32(352)
    for g, w in zip(got3, d3[0, :n3].cpu().numpy()):
        assert [int(v) for v in g[:4]] == [int(np.float32(w[0]) * sw), int(np.float32(w[1]) * sh), int(np.float32(w[2]) * sw), int(np.float32(w[3]) * sh)]
        assert int(g[4]) == int(w[5]) and np.float32(float(g[5])) == np.float32(w[4])

    # VERDICT r04 missing 6: a model whose activations leave the fp16x3 plan's range must come back from detection() as an
    # ERROR (YFV2_ERR_RANGE), not as boxes - the C++ class waits for its results anyway and polls the guard there
    deep = {k: v.clone() for k, v in yfv2.random_state_dict(3).items()}
    deep["backbone.stage4.3.branch_main.6.weight"] *= 3000.0
    deep["backbone.stage4.3.branch_main.6.bias"] *= 3000.0
    dpath = str(tmp_path / "deep.yfv2w")
    assert yfv2.export_weights(deep, dpath) > 0
    ipath = str(tmp_path / "img_guard.raw")
    hwc.tofile(ipath)
    r = subprocess.run([exe, dpath, anchors, ipath, "352", "352", "0.3", "0.4"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and "YFV2_BF6=0" in r.stderr and r.stdout.strip() == "", (r.returncode, r.stdout[:200], r.stderr[:500])

    up = np.ascontiguousarray(hwc.repeat(2, axis=0).repeat(2, axis=1))
    got2 = run(up, 704, 704)   # every output pixel is the rounded mean of four equal pixels: the resized image is identical
    assert len(got2) == n
    for g, w in zip(got2, want):
        assert [int(v) for v in g[:4]] == [int(np.float32(v) * np.float32(2.0)) for v in w[:4]]
        assert int(g[4]) == int(w[5])
        assert np.float32(float(g[5])) == np.float32(w[4])


def test_evaluation_loop_matches_oracle_pipeline(yfv2, model, dev, cfg, images_u8, coco_weights):
    """SURVEY.md 8(f) row 2, the whole loop: evaluation() (utils/utils.py:360-397 signature) over a two-batch loader of
    the reference images with targets derived from the oracle's own detections, against the same loop composed from
    the oracle (forward/decode/NMS at conf 0.01, get_batch_statistics restatement) and ap_per_class (pinned bit-exact
    against the reference on CPU).  A detection whose confidence sits within fp32 noise of 0.01 may fall on either side,
    so the four means are compared to 5e-3, not bitwise."""
    imgs = torch.from_numpy(images_u8[:6])
    x = imgs.float() / 255.0
    _, _, (rows03, _) = oracle.detect(coco_weights, x, cfg["anchors"], cfg["height"], 0.3, 0.4)
    rng = np.random.default_rng(5)
    W, H = float(cfg["width"]), float(cfg["height"])

    def targets_for(lo, hi):
        t = []
        for b in range(lo, hi):
            for r in rows03[b]:
                bx = r[:4] + rng.normal(0, 3.0, 4).astype(np.float32)
                t.append([b - lo, r[5], (bx[0] + bx[2]) / 2 / W, (bx[1] + bx[3]) / 2 / H, (bx[2] - bx[0]) / W, (bx[3] - bx[1]) / H])
            t.append([b - lo, 79.0, 0.1, 0.1, 0.05, 0.05])      # a ground-truth object nobody finds
        return torch.tensor(np.asarray(t, np.float32))

    loader = [(imgs[0:4], targets_for(0, 4)), (imgs[4:6], targets_for(4, 6))]
    got = yfv2.evaluation(loader, cfg, model, dev)
    assert got is not None and len(got) == 4

    tps, confs, clss, labels = [], [], [], []
    for bi, tg in loader:
        _, _, (rows, _) = oracle.detect(coco_weights, bi.float() / 255.0, cfg["anchors"], cfg["height"], 0.01, 0.4)
        t = tg.numpy().copy()
        labels += t[:, 1].tolist()
        c = t[:, 2:].copy()
        t[:, 2], t[:, 3] = c[:, 0] - c[:, 2] / np.float32(2), c[:, 1] - c[:, 3] / np.float32(2)
        t[:, 4], t[:, 5] = c[:, 0] + c[:, 2] / np.float32(2), c[:, 1] + c[:, 3] / np.float32(2)
        t[:, 2:] *= np.asarray([W, H, W, H], np.float32)
        for tp, sc, lb in oracle.get_batch_statistics(rows, t, 0.5):
            tps.append(tp); confs.append(sc); clss.append(lb)
    want = yfv2.ap_per_class(np.concatenate(tps), np.concatenate(confs), np.concatenate(clss), labels)
    assert want[2] > 0.2, "the synthetic targets should be found: mean AP %g" % want[2]
    assert np.allclose(np.asarray(got), np.asarray(want), atol=5e-3), (got, want)
    assert yfv2.evaluation([], cfg, model, dev) is None


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 480, 640), (2, 120, 160), (1, 352, 352), (1, 1080, 1920), (2, 353, 351), (1, 1, 1), (1, 704, 704)])
def test_resize_u8_bit_exact_vs_oracle(yfv2, model, dev, shape):
    """SURVEY.md 8(f) row 1, the resize half: yfv2_resize_u8 against the oracle's restatement of cv2.resize(INTER_LINEAR)
    for uint8 (bit-exact: integer arithmetic after the float32 coefficient tables).  Reductions, enlargements, odd sizes
    whose rows start at every byte alignment, a one-pixel frame, and a source buffer that itself starts at an odd address."""
    B, sh, sw = shape
    rng = np.random.default_rng(sh * 31 + sw)
    frames = (rng.random((B, sh, sw, 3)) * 255).astype(np.uint8)
    want = oracle.resize_linear_u8(frames, 352, 352)
    eng = yfv2.get_engine(dev, 352, 352)
    got = eng.resize(torch.from_numpy(frames).to(dev))
    assert got.dtype == torch.uint8 and tuple(got.shape) == (B, 352, 352, 3)
    assert np.array_equal(got.cpu().numpy(), want)
    flat = torch.zeros(frames.size + 1, dtype=torch.uint8, device=dev)
    flat[1:] = torch.from_numpy(frames).to(dev).reshape(-1)
    got2 = eng.resize(flat[1:].view(B, sh, sw, 3))      # base pointer = allocation + 1
    assert np.array_equal(got2.cpu().numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("classes,hw", [(100, (416, 416)), (80, (512, 512)), (80, (640, 384)), (20, (96, 1024)), (255, (384, 384))],
                         ids=["416x416-100-classes", "512x512", "640x384", "96x1024", "384x384-255-classes"])
def test_sizes_and_class_counts_beyond_the_former_static_limits(classes, hw):
    """VERDICT r01-r03 carry-over: width <= 384, decode rows <= 2048 and classes <= 93 were static bounds; the reference sizes
    everything from its .data file (utils/utils.py:13-65,303-306,332).  Now: any width, up to 4096 decode rows (512x512,
    640x384), up to 255 classes (the class head in slices of 96 output channels, decode with 64 class slots per lane, decode
    + NMS as two launches, four sort keys per thread above 2048 rows).  Same script and same checks as the class-count cases:
    logits within the noise floor, decode, bit-exact NMS, detect == three calls, an all-rows-are-candidates NMS stress."""
    import subprocess
    import sys

    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_cases", "class_counts.py")
    r = subprocess.run([sys.executable, script, str(classes), str(hw[0]), str(hw[1])], capture_output=True, text=True, timeout=400)
    marks = [ln for ln in r.stdout.splitlines() if ln.startswith("[class_counts]")]
    assert r.returncode == 0, ("exit code %d after %r" % (r.returncode, marks[-1] if marks else "no marker"), r.stdout[-1500:], r.stderr[-3000:])
    assert "PARITY OK classes=%d" % classes in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("classes", [1, 2, 3, 5, 6, 9, 20, 93, 94, 100, 255])
def test_other_class_counts_in_their_own_interpreter(classes):
    """The reference is trained on custom `.data` files (classes is free, utils/utils.py:13-65, model/detector.py:17-19).
    Every class count whose softmax quarters are uneven or empty (1, 2, 3, 5, 6, 9: ceil(nc/4)*3 >= nc, the case whose masked
    loads once read past the class tensor), 20 (two tiles of the chained cls-tower conv) and the maximum 93, against the
    oracle, and 94 / 100 / 255 (more than one chained output conv holds: the class head runs in slices): logits, decode, bit-exact
    NMS, fused detect (tests/gpu_cases/class_counts.py).  Each case runs in its own
    interpreter so that a device fault cannot take the rest of the suite with it - but a crash, a hang or a mismatch FAILS."""
    import subprocess
    import sys

    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_cases", "class_counts.py")
    r = subprocess.run([sys.executable, script, str(classes)], capture_output=True, text=True, timeout=240)
    marks = [ln for ln in r.stdout.splitlines() if ln.startswith("[class_counts]")]
    assert r.returncode == 0, ("exit code %d after %r" % (r.returncode, marks[-1] if marks else "no marker"), r.stdout[-1500:], r.stderr[-3000:])
    assert "PARITY OK classes=%d" % classes in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("B,kind,hw", [(6144, "fp32", (352, 352)), (4096, "uint8", (352, 352)), (2304, "fp32", (512, 512))],
                         ids=["6144-fp32-9GiB-input", "4096-uint8", "2304-fp32-512x512"])
def test_batches_beyond_4_gib_and_2_31_elements(B, kind, hw):
    """Maximum sizes: one call on 6144 images (input 9.1 GiB = 2.28e9 elements, stem output 4.6 GiB, stage-2 planes 4.6 GiB, decoded
    tensor 3.8 GiB: every tensor that can pass 2^31 elements or 4 GiB does), on 4096 uint8 images and on 2304 images of 512x512 (the
    general-size plan; round 5's probe also ran YFV2_BF6=0 / YFV2_FUSED=0 / YFV2_POSTFUSE=0, 640x384 and 288x384 uint8: tools/gpu_r5_probe2.sh).  The reference has no
    batch bound (utils/utils.py:251 loops over whatever it is given); a 288 GB device holds these.  Property checked
    (tests/gpu_cases/large_batch.py): the batch is K copies of one 256-image block and every copy's logits, decoded rows and
    detections are BIT-identical to the block's own batch-of-256 result - a per-image base address computed in 32 bits wraps
    inside such a batch and fails it.  Own interpreter: a fault must not take the suite down, but it FAILS."""
    import subprocess
    import sys

    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_cases", "large_batch.py")
    r = subprocess.run([sys.executable, script, str(B), kind, str(hw[0]), str(hw[1])], capture_output=True, text=True, timeout=400)
    marks = [ln for ln in r.stdout.splitlines() if ln.startswith("[large_batch]")]
    assert r.returncode == 0, ("exit code %d after %r" % (r.returncode, marks[-1] if marks else "no marker"), r.stdout[-1500:], r.stderr[-3000:])
    assert "LARGE BATCH OK B=%d" % B in r.stdout


def test_detect_pipeline_matches_one_handle_bit_for_bit(yfv2, dev, coco_weights, images_u8, cfg):
    """DetectPipeline (bench.py's `value` loop as a product class): seven different batches - fp32 and uint8, full and partial -
    rotating over three handles / streams give exactly what one handle gives batch by batch; a ticket whose slot was reused
    is refused; results arrive on the caller's stream."""
    pipe = yfv2.DetectPipeline(dev, 352, 352, 80, 3, anchors=cfg["anchors"], max_batch=12, depth=3)
    pipe.load_state_dict(coco_weights)
    one = yfv2.Engine(dev, 352, 352, 80, 3, anchors=cfg["anchors"], max_batch=12)
    one.load_state_dict(coco_weights)
    batches = []
    for k in range(7):
        xb = _batch_from_reference_images(images_u8, 12 if k % 3 else 7, seed=40 + k)
        if k % 2:
            xb = (xb * 255.0).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()   # the uint8 HWC entry
        batches.append(xb.to(dev))
    want = [tuple(t.clone() for t in one.detect(x, 0.25, 0.4)) for x in batches]
    tickets, got = [], []
    for k, x in enumerate(batches):
        tickets.append(pipe.submit(x, 0.25, 0.4))
        if k >= 2:                                  # consume with two batches still in flight
            got.append(tuple(t.clone() for t in pipe.result(tickets[k - 2])))
    got += [tuple(t.clone() for t in pipe.result(t)) for t in tickets[-2:]]
    torch.cuda.synchronize()
    assert sum(int(w[2].sum()) for w in want) > 50
    for k, (w, g) in enumerate(zip(want, got)):
        assert torch.equal(w[2], g[2]), k
        for b in range(w[2].shape[0]):
            n = int(w[2][b])
            assert torch.equal(w[0][b, :n], g[0][b, :n]) and torch.equal(w[1][b, :n], g[1][b, :n]), (k, b)
    with pytest.raises(RuntimeError):
        pipe.result(tickets[0])                     # its slot now holds batch 6
    d, i, c = pipe.result(tickets[6], host=True)
    assert torch.equal(c, want[6][2])


def _engine_with_env(yfv2, dev, env, **kw):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return yfv2.Engine(dev, 352, 352, 80, 3, **kw)     # plan switches and the lane count are read when the handle is created
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("hw,B", [((352, 352), 9), ((320, 320), 3), ((288, 384), 2), ((64, 96), 3), ((32, 32), 4), ((352, 32), 2), ((96, 1024), 2),
                                  ((512, 512), 2), ((416, 416), 2), ((352, 352), 300)])
def test_front_kernel_is_bit_identical_to_stem_plus_stage2_0(yfv2, dev, hw, B):
    """Round 5: the stem and stage2.0 as ONE launch (front_kernel: a lane owns two adjacent pooled columns, the stem's matrix-core
    output layout IS the stride-2 block's input layout, the [H/4][W/4][24] tensor never leaves the registers) against the two
    launches it replaces (YFV2_FRONT=0).  Every value passes through the same instructions on the same operands in the same
    order, so stage 2 and the six logit maps must be BIT-identical - at sizes whose strips and bands are ragged (64x96, 32x32,
    352x32, 96x1024), at the general sizes, and with more images than compute units.  The stem's own output, which the fused
    launch never writes, is still there for the debug hook (re-run from the last input) and equal too.  uint8 (B,H,W,3) input runs
    the same fusion on stem_h3u_kernel's arithmetic (front2_kernel<.., U8>: a lane's two pooled columns are 24 consecutive bytes)
    and is compared the same way: logits and stage 2 bit-identical to stem_h3u_kernel + s2h_kernel."""
    H, W = hw
    sd = yfv2.random_state_dict(17)
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(H * 7 + W + B)).to(dev)
    outs = []
    for env in ({"YFV2_FRONT": "0"}, {}):
        old = os.environ.get("YFV2_FRONT")
        os.environ.update(env)
        try:
            e = yfv2.Engine(dev, H, W, 80, 3, max_batch=B)
        finally:
            if old is None:
                os.environ.pop("YFV2_FRONT", None)
            else:
                os.environ["YFV2_FRONT"] = old
        e.load_state_dict(sd)
        names = [s["name"] for s in e.stages()]
        if env:
            px_plan = "lane-per-pixel" in names[1]      # maps too small for the streaming stage-2 kernels run layer-wise blocks: nothing to fuse there
        assert names[0].startswith("stem + backbone.stage2.0 in one launch") == (not env and px_plan), names[:2]
        logits = [t.clone() for t in e.forward(x)]
        acts = [e.debug_activation(w, min(B, 9)) for w in (0, 1)]
        xu = (x[:min(B, 9)].permute(0, 2, 3, 1) * 255.0).round().clamp(0, 255).to(torch.uint8).contiguous()
        lu = [t.clone() for t in e.forward(xu)] + [e.debug_activation(w, xu.shape[0]) for w in (0, 1)]
        e.check_finite("front / two-launch plan")
        outs.append((logits, acts, lu))
    (l0, a0, u0), (l1, a1, u1) = outs
    for k, (p, q) in enumerate(zip(a0, a1)):
        assert torch.equal(p, q), "%s: %d of %d elements differ" % (("stem output (debug hook)", "stage 2")[k], int((p != q).sum()), p.numel())
    for key, p, q in zip(LOGIT_KEYS, l0, l1):
        assert torch.equal(p, q), "%s differs between the one-launch and the two-launch front" % key
    for key, p, q in zip(list(LOGIT_KEYS) + ["stem output (debug hook)", "stage 2"], u0, u1):
        assert torch.equal(p, q), "uint8 input, %s" % key


@pytest.mark.parametrize("lanes", [2, 3])
def test_lanes_inside_one_call_are_bit_identical(yfv2, dev, coco_weights, images_u8, cfg, lanes):
    """YFV2_LANES=N: one yfv2_detect / yfv2_forward call cuts its batch into N slices on N internal
    streams, forked from and joined into the caller's stream.  Images are independent, so every output - logits, detections,
    indices, counts, the debug activations - equals the unsliced handle's bit for bit: full batch, a batch that does not divide
    evenly, one below the lane threshold (runs unsliced), fp32 and uint8 entry; and work enqueued by the caller right after
    the call sees the finished result (the join)."""
    B = 32 * lanes + 7
    one = _engine_with_env(yfv2, dev, {"YFV2_LANES": "1"}, anchors=cfg["anchors"], max_batch=B)
    many = _engine_with_env(yfv2, dev, {"YFV2_LANES": str(lanes)}, anchors=cfg["anchors"], max_batch=B)
    one.load_state_dict(coco_weights)
    many.load_state_dict(coco_weights)
    x = _batch_from_reference_images(images_u8, B, seed=77).to(dev)
    xu = (x * 255.0).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    side = torch.cuda.Stream(device=dev)
    for inp in (x, xu, x[: 32 * lanes], x[:20]):
        n = inp.shape[0]
        want_logits = [t.clone() for t in one.forward(inp)]
        want = [t.clone() for t in one.detect(inp, 0.25, 0.4)]
        with torch.cuda.stream(side):                      # a caller stream that is not the NULL stream
            side.wait_stream(torch.cuda.current_stream(dev))
            got_logits = many.forward(inp)
            total = sum(t.double().sum() for t in got_logits)      # consumer enqueued right behind the call, same stream
            acts = [many.debug_activation(w, n) for w in (0, 1, 2, 5)] if inp is x else None
            got = many.detect(inp, 0.25, 0.4)
            cnt_sum = got[2].sum()
        side.synchronize()
        for a, b in zip(want_logits, got_logits):
            assert torch.equal(a, b)
        assert float(total) == float(sum(t.double().sum() for t in want_logits))
        assert torch.equal(want[2], got[2]) and int(cnt_sum) == int(want[2].sum()) and int(cnt_sum) > n // 2
        for b in range(n):
            k = int(want[2][b])
            assert torch.equal(want[0][b, :k], got[0][b, :k]) and torch.equal(want[1][b, :k], got[1][b, :k]), b
        if acts is not None:
            one.forward(inp)
            for w, a in zip((0, 1, 2, 5), acts):
                assert torch.equal(one.debug_activation(w, n), a), w


def test_range_guard_of_the_fp16x3_plan(yfv2, dev, cfg):
    """VERDICT r03 weak 3 / ADVICE medium: the default plan's fp16x3 contractions are valid for |activation| < 4094 (fp32 input
    |x| < 255.9); beyond that an operand splits into (+Inf, -Inf), the products are NaN and the ReLU behind the conv would turn
    them into a silent 0.  Every kernel of the plan checks its matrix-core accumulators before the ReLU and sets a sticky word
    (include/yfv2.h yfv2_nonfinite).  (1) ordinary weights and inputs: never set; (2) a checkpoint whose stem BatchNorm gain
    puts stage-2 activations near 1e4: set, cleared by the query, and the SAME weights on a YFV2_BF6=0 handle (fp32 matrix
    instructions, no bound) stay within the noise floor of the oracle and never set it; (3) an fp32 input beyond 255.9, a NaN
    pixel, an Inf pixel: set; (4) through the reference surface: handel_preds raises instead of returning zeros."""
    w = yfv2.random_state_dict(3)
    g = torch.Generator().manual_seed(21)
    x = torch.rand(3, 3, 352, 352, generator=g)
    eng = yfv2.Engine(dev, 352, 352, 80, 3, anchors=cfg["anchors"], max_batch=4)
    eng.load_state_dict(w)
    eng.forward(x.to(dev))
    assert not eng.nonfinite()
    eng.forward((x * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().to(dev))
    assert not eng.nonfinite()
    # (2) large activations
    big = {k: v.clone() for k, v in w.items()}
    big["backbone.first_conv.1.weight"] *= 4000.0
    big["backbone.first_conv.1.bias"] *= 4000.0
    ref_stem = oracle.forward_stages(big, x)["stem"]
    assert float(ref_stem.abs().max()) > 4094.0 * 1.5, float(ref_stem.abs().max())
    eng.load_state_dict(big)
    eng.forward(x.to(dev))
    assert eng.nonfinite(), "activations of %.0f went through the fp16x3 plan unnoticed" % float(ref_stem.abs().max())
    assert not eng.nonfinite()                               # the query clears the word
    fp32 = _engine_with_env(yfv2, dev, {"YFV2_BF6": "0"}, anchors=cfg["anchors"], max_batch=4)
    fp32.load_state_dict(big)
    got = fp32.forward(x.to(dev))
    assert not fp32.nonfinite()
    assert all(torch.isfinite(t).all() for t in got)
    _assert_logits_within_noise_floor(got, big, x, "YFV2_BF6=0 on large activations")
    # (3) inputs outside the contract
    eng.load_state_dict(w)
    for bad in (x * 300.0, x.clone().index_put_((torch.tensor(1), torch.tensor(2), torch.tensor(100), torch.tensor(200)), torch.tensor(float("nan"))),
                x.clone().index_put_((torch.tensor(2), torch.tensor(0), torch.tensor(351), torch.tensor(0)), torch.tensor(float("inf")))):
        eng.forward(x.to(dev))
        assert not eng.nonfinite()
        eng.forward(bad.to(dev))
        assert eng.nonfinite()
    # deep in the network too: a gain on the LAST backbone block's BatchNorm (stage-4 chain -> FPN reduce -> towers)
    deep = {k: v.clone() for k, v in w.items()}
    deep["backbone.stage4.3.branch_main.6.weight"] *= 3000.0
    deep["backbone.stage4.3.branch_main.6.bias"] *= 3000.0
    eng.load_state_dict(deep)
    eng.detect(x.to(dev), 0.3, 0.4)
    torch.cuda.synchronize(dev)
    assert eng.peek_nonfinite() and eng.peek_nonfinite()     # the look neither waits nor clears ...
    with pytest.raises(yfv2.Yfv2Error, match="YFV2_BF6=0"):   # ... and the NEXT detect on the handle refuses to go on silently
        eng.detect(x.to(dev), 0.3, 0.4)
    assert not eng.peek_nonfinite()                           # raising cleared the word
    eng.detect(x.to(dev), 0.3, 0.4, check=False)              # opt-out: enqueue regardless
    assert eng.nonfinite() and not eng.nonfinite()
    # (4) the reference surface
    m = yfv2.Detector(80, 3, True).to(dev)
    m.load_state_dict(big)
    m.eval()
    with pytest.raises(yfv2.Yfv2Error, match="YFV2_BF6=0"):
        yfv2.handel_preds(m(x.to(dev)), cfg, dev)
    m.load_state_dict(w)
    assert tuple(yfv2.handel_preds(m(x.to(dev)), cfg, dev).shape) == (3, 1815, 85)


def test_two_handles_on_two_streams_stay_bit_identical(yfv2, dev, coco_weights, images_u8, cfg):
    """Two independent handles working at the same time on two streams (what DetectPipeline and the lanes do at batch sizes
    that fill the machine) must each give what they give alone - uint8 and fp32 input, twelve rounds.  Round 4 found the
    uint8 stem's 16-byte buffer stores corrupted in 4-5 of 12 such rounds (a store-data WAR hazard hipcc does not cover when
    the store's soffset is a register: yfv2_internal.h, yfv2_after_wide_buffer_store); one kernel at a time never showed it."""
    B = 70
    one = yfv2.Engine(dev, 352, 352, 80, 3, anchors=cfg["anchors"], max_batch=B)
    two = yfv2.Engine(dev, 352, 352, 80, 3, anchors=cfg["anchors"], max_batch=B)
    one.load_state_dict(coco_weights); two.load_state_dict(coco_weights)
    x = _batch_from_reference_images(images_u8, B, seed=77).to(dev)
    xu = (x * 255.0).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    for src in (xu, x):
        torch.cuda.synchronize(dev)
        full = [t.clone() for t in one.forward(src)]
        for rep in range(12):
            torch.cuda.synchronize(dev)
            with torch.cuda.stream(s1):
                a = one.forward(src[:35])
            with torch.cuda.stream(s2):
                b = two.forward(src[35:])
            torch.cuda.synchronize(dev)
            for f, p, q in zip(full, a, b):
                assert torch.equal(f[:35], p) and torch.equal(f[35:], q), (str(src.dtype), rep)
