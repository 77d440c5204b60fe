"""Pin the CPU oracle (oracle/yfv2_oracle.py) against outputs of the reference
itself (tests/golden/*.npz, produced by tests/golden/make_golden.py importing
/root/reference).  CPU only.

Tolerances: the goldens were produced single-threaded; a different thread count
or CPU ISA changes oneDNN's summation order, so logits are compared at 2e-5
(measured reference-vs-reference noise is <= 1.7e-5) rather than
bit-exactly.  Decode from *golden logits* is <= 2 ulp (torch's strided-vs-
contiguous sigmoid).  NMS from *golden decoded* is bit-exact.
"""
import numpy as np
import torch

from conftest import unpack_ragged
from oracle import yfv2_oracle as oracle

LOGIT_KEYS = ("reg2", "obj2", "cls2", "reg3", "obj3", "cls3")


def _logits(z):
    return [torch.from_numpy(z["logit_" + k]) for k in LOGIT_KEYS]


def test_forward_real_images(golden_real, images_u8, coco_weights):
    x = torch.from_numpy(images_u8).float() / 255.0
    assert abs(x.double().sum().item() - float(golden_real["x_sum64"])) < 1e-6
    preds = oracle.forward(coco_weights, x)
    for p, k in zip(preds, LOGIT_KEYS):
        assert p.shape == golden_real["logit_" + k].shape
        np.testing.assert_allclose(p.numpy(), golden_real["logit_" + k], rtol=0, atol=2e-5)


def test_forward_seeded_rand(golden_rand, coco_weights):
    torch.manual_seed(1234)
    x = torch.rand(2, 3, 352, 352)
    np.testing.assert_array_equal(x.flatten()[::100003].numpy(), golden_rand["x_probe"])
    preds = oracle.forward(coco_weights, x)
    for p, k in zip(preds, LOGIT_KEYS):
        np.testing.assert_allclose(p.numpy(), golden_rand["logit_" + k], rtol=0, atol=2e-5)


def _ulp_close(a, b, ulps):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    tol = ulps * np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32))
    return np.all(np.abs(a.astype(np.float64) - b.astype(np.float64)) <= tol + 1e-12)


def test_decode_from_golden_logits(golden_real, golden_rand, golden_kat, cfg):
    for z in (golden_real, golden_rand, golden_kat):
        dec = oracle.decode(_logits(z), cfg["anchors"], cfg["height"])
        assert dec.shape == z["decoded"].shape and dec.dtype == np.float32
        assert _ulp_close(dec, z["decoded"], 2)


def test_decode_known_answers(golden_kat):
    """Row order (y,x,anchor), grid, stride and anchors: SURVEY.md 8(c) row 3."""
    dec = golden_kat["decoded"][0]
    r0, r1 = golden_kat["rows"]
    assert (r0, r1) == (352, 1565)
    np.testing.assert_allclose(dec[352, :4], [120.0, 88.0, 37.88, 51.48], rtol=1e-6)
    np.testing.assert_allclose(dec[1565, :4], [144.0, 112.0, 279.92, 258.87], rtol=1e-6)
    assert dec[352, 5:].argmax() == 17 and dec[1565, 5:].argmax() == 63


def _check_nms(z, prefix, conf, iou):
    rows, idx = oracle.non_max_suppression(z["decoded"], conf, iou)
    g_rows, g_idx = unpack_ragged(z, prefix)
    assert len(rows) == len(g_rows)
    for b in range(len(rows)):
        assert rows[b].shape == g_rows[b].shape, (prefix, b)
        assert np.array_equal(rows[b].view(np.uint32), g_rows[b].view(np.uint32)), (prefix, b)
        assert np.array_equal(idx[b], g_idx[b]), (prefix, b)


def test_nms_bit_exact_real(golden_real, golden_rand):
    for z in (golden_real, golden_rand):
        _check_nms(z, "nms_03_04", 0.3, 0.4)
        _check_nms(z, "nms_001_04", 0.01, 0.4)
        _check_nms(z, "nms_03_045", 0.3, 0.45)


def test_nms_bit_exact_stress(golden_stress):
    _check_nms(golden_stress, "nms_03_04", 0.3, 0.4)
    _check_nms(golden_stress, "nms_001_04", 0.01, 0.4)
    _check_nms(golden_stress, "nms_025_06", 0.25, 0.6)
    # the stress set must actually hit max_det and the tie-break path
    assert golden_stress["nms_03_04_count"].max() == 300


def test_survivor_indices_match_survey(golden_real):
    """SURVEY.md 'Sanity goldens': img/000139.jpg -> rows 491, 1662, 1061."""
    _, idx = unpack_ragged(golden_real, "nms_03_04")
    assert list(idx[1]) == [491, 1662, 1061]
    assert list(idx[2]) == [1670]


def test_nms_edge_cases():
    empty = np.zeros((2, 1815, 85), np.float32)
    rows, idx = oracle.non_max_suppression(empty, 0.3, 0.4)
    assert all(r.shape == (0, 6) for r in rows) and all(i.shape == (0,) for i in idx)
    # obj passes, conf does not
    one = empty.copy(); one[0, 7, :5] = [10, 10, 4, 4, 0.9]; one[0, 7, 5] = 0.2
    rows, _ = oracle.non_max_suppression(one, 0.3, 0.4)
    assert rows[0].shape == (0, 6)
    # class filter (utils.py:271-272)
    one[0, 7, 5 + 3] = 0.9
    rows, idx = oracle.non_max_suppression(one, 0.3, 0.4, classes=[3])
    assert rows[0].shape == (1, 6) and idx[0][0] == 7 and rows[0][0, 5] == 3.0
    rows, _ = oracle.non_max_suppression(one, 0.3, 0.4, classes=[4])
    assert rows[0].shape == (0, 6)


def test_random_weights_are_complete(coco_weights):
    from yolo_fastestv2_amd.weights import random_state_dict
    rw = random_state_dict(0)
    assert set(rw) == set(coco_weights)
    for k in rw:
        assert tuple(rw[k].shape) == tuple(coco_weights[k].shape), k
    preds = oracle.forward(rw, torch.rand(1, 3, 352, 352))
    assert all(torch.isfinite(p).all() for p in preds)
    assert float(preds[2].abs().max()) > 1e-3  # activations neither vanish nor explode
    assert float(preds[2].abs().max()) < 1e3


def test_oracle_batch_statistics_match_reference_golden(golden_stats):
    """utils.py:194-230 restated (oracle.get_batch_statistics) vs the flags the reference function produced."""
    dets, _ = unpack_ragged(golden_stats, "dets")
    for thr, key in ((0.5, "tp_050"), (0.75, "tp_075")):
        got = oracle.get_batch_statistics(dets, golden_stats["targets"], thr)
        flat = np.concatenate([t for t, _, _ in got])
        assert np.array_equal(flat.astype(np.uint8), golden_stats[key])
        assert [t.shape[0] for t, _, _ in got] == golden_stats[key + "_count"].tolist()
    # the semantics the fixture was built to exercise
    got = oracle.get_batch_statistics(dets, golden_stats["targets"], 0.5)
    assert got[5][0].sum() == 0 and got[5][0].shape[0] == dets[5].shape[0]       # image without targets: all false positives
    assert oracle.get_batch_statistics([None, dets[1]], golden_stats["targets"], 0.5).__len__() == 1   # None outputs are skipped


def test_oracle_resize_properties():
    """oracle.resize_linear_u8 restates cv2.resize(INTER_LINEAR) for uint8 from the published algorithm ("parity unpinned":
    OpenCV is absent from the image).  What CAN be checked here: the equal-size copy, an exact 2x reduction equals the
    rounded 2x2 mean (11-bit weights 1024/1024; also what OpenCV's INTER_AREA shortcut for that case computes), a
    constant frame stays constant, and an enlargement agrees with PIL's bilinear (same sampling positions, different
    fixed-point rounding) to one grey level."""
    rng = np.random.default_rng(0)
    img = (rng.random((120, 160, 3)) * 255).astype(np.uint8)
    assert np.array_equal(oracle.resize_linear_u8(img, 160, 120), img)
    big = (rng.random((704, 704, 3)) * 255).astype(np.uint8)
    mean4 = ((big[0::2, 0::2].astype(int) + big[0::2, 1::2] + big[1::2, 0::2] + big[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    assert np.array_equal(oracle.resize_linear_u8(big, 352, 352), mean4)
    assert np.unique(oracle.resize_linear_u8(np.full((480, 640, 3), 137, np.uint8), 352, 352)).tolist() == [137]
    out = oracle.resize_linear_u8(np.stack([img, img[::-1]]), 352, 288)
    assert out.shape == (2, 288, 352, 3) and out.dtype == np.uint8
    try:
        from PIL import Image
    except ImportError:
        return
    pil = np.asarray(Image.fromarray(img).resize((352, 352), Image.BILINEAR))
    assert np.abs(oracle.resize_linear_u8(img, 352, 352).astype(int) - pil.astype(int)).max() <= 1


def test_noise_floor_golden_and_the_rule_built_on_it(golden_floor):
    """golden_floor.npz (make_golden.py floor: the reference's modules in fp32 and in float64 on the bench workload).  On the
    CPU: this host's float64 oracle reproduces the reference's float64 logits; the oracle's fp32 forward + decode sits at the
    recorded floor; the noise-floor check of tests/test_gpu_parity.py accepts the oracle's own execution (with its survivors)
    and REJECTS an execution whose logits carry three floors of extra error - the check can fail."""
    import bench
    import pytest
    import test_gpu_parity as tg
    from conftest import floor_inputs
    sd, x = floor_inputs(golden_floor)
    keep, n = int(golden_floor["keep"]), 4
    xs = x[:n]
    p64 = oracle.forward64(sd, xs)
    for k, t in zip(tg.LOGIT_KEYS, p64):
        assert t.dtype == torch.float64
        assert np.abs(t[:keep].numpy() - golden_floor["logit64_" + k]).max() <= 1e-9, k
    p32 = oracle.forward(sd, xs)
    for k, a, b in zip(tg.LOGIT_KEYS, p32, p64):
        e_max, e_rms = tg._err_stats(a.numpy(), b.numpy())
        assert e_max <= 1.5 * golden_floor["err_" + k][0] and 0.5 * golden_floor["err_" + k][1] <= e_rms <= 1.5 * golden_floor["err_" + k][1], (k, e_max, e_rms)
    assert max(float(golden_floor["err_" + k][0]) for k in tg.LOGIT_KEYS) < 1e-4      # the reference itself is inside 1e-4 absolute at logit scale 36
    err_ref = {k: golden_floor["err_" + k] for k in tg.LOGIT_KEYS}
    err_ref["decoded"] = golden_floor["err_decoded"]
    o_dec = oracle.decode(p32, bench.ANCHORS, 352)
    _, o_idx = oracle.non_max_suppression(o_dec, 0.3, 0.4)
    fl = tg._bench_regime_floor_check(sd, xs, list(p32), o_dec, [[int(v) for v in i] for i in o_idx], err_ref)
    assert not fl["violations"] and fl["record"]["end_to_end_n_diff"] == 0
    assert fl["record"]["worst_logit_ratio_device_over_reference"] <= 1.5
    g = torch.Generator().manual_seed(3)
    noisy = [t + 3 * float(golden_floor["err_" + k][1]) * torch.randn(t.shape, generator=g) for k, t in zip(tg.LOGIT_KEYS, p32)]
    with pytest.raises(AssertionError):
        tg._bench_regime_floor_check(sd, xs, noisy, o_dec, [[int(v) for v in i] for i in o_idx], err_ref)
