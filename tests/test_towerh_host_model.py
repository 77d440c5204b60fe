"""towerh_kernel (yfv2_towerh.hip: the FPN towers - depthwise 5x5 with wave-uniform taps, fp16x3 pointwise conv, chained
output conv) pinned on the CPU: the image the host packs for it (yfv2_debug_plan_image; the kernel's image follows
tower2_kernel's in the blob) is decoded - two-term fp16 filters x 2^sw, tap table with the BN constants x 2^4, BN scale
carrying 2^-(sw+4), bias, output-conv unscale - and a numpy model of the kernel's ARITHMETIC built only from it (fp32
depthwise + BN + ReLU on the x16 scale, split into two fp16 terms, three exact products per MAC, power-of-two unscales)
must reproduce the oracle's tower half and output conv."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import yolo_fastestv2_amd as yfv2
from yolo_fastestv2_amd import _lib
from yolo_fastestv2_amd._lib import Config, TensorDesc

WP_FL, CS_FL, TILE_FL, TAPS_FL = 6400, 384, 1280, 5 * 4 * 27 * 4 + 16


def _images(w, classes):
    host = {k: v.float().contiguous() for k, v in w.items() if v.is_floating_point()}
    arr = (TensorDesc * len(host))()
    for i, (k, t) in enumerate(host.items()):
        arr[i].name, arr[i].data, arr[i].numel = k.encode(), t.data_ptr(), t.numel()
    cfg = Config()
    cfg.classes, cfg.anchor_num, cfg.height, cfg.width, cfg.max_batch, cfg.device = classes, 3, 352, 352, 1, 0
    L = _lib.lib()
    ns, nb = C.c_int32(0), C.c_int64(0)
    assert L.yfv2_debug_plan_dryrun(C.byref(cfg), arr, len(host), C.byref(ns), C.byref(nb)) == 0
    out = {}
    cap = 2 * (WP_FL + 6 * TILE_FL + 2000 + 480) + CS_FL + TAPS_FL + 64
    buf = np.zeros(cap, np.float32)
    nm = C.create_string_buffer(256)
    n_view = L.yfv2_debug_plan_image(C.byref(cfg), arr, len(host), -1, None, 0, None, 0)    # steps of the image view (yfv2.h: the stem and stage2.0
    assert ns.value <= n_view <= ns.value + 1                                                # are two steps there even where they are one launch)
    for st in range(n_view):
        n = L.yfv2_debug_plan_image(C.byref(cfg), arr, len(host), st, nm, 256, buf.ctypes.data_as(C.c_void_p), cap)
        assert n >= 0, st
        out[nm.value.decode()] = buf[:max(n, 0)].copy()
        for job in range(4):                                                    # launches that run several tower halves: each half's own image
            n = L.yfv2_debug_plan_image(C.byref(cfg), arr, len(host), st + 1000 * (job + 1), nm, 256, buf.ctypes.data_as(C.c_void_p), cap)
            if n < 0:
                break
            out[nm.value.decode()] = buf[:n].copy()
    return out


def _two_term(fr, tiles):
    """[tile][chunk 5][64 lanes][4 dwords: term 1, term 1, term 2, term 2] -> (16 tiles, 80) float64 = sum of the two fp16 terms"""
    u = fr.view(np.uint32).reshape(tiles, 5, 64, 4)
    halves = np.stack((u & 0xFFFF, u >> 16), -1).astype(np.uint16).view(np.float16).astype(np.float64)   # [...][dword][element]
    m = np.zeros((16 * tiles, 80))
    for t in range(tiles):
        for s in range(5):
            for l in range(64):
                r, c0 = 16 * t + (l & 15), 16 * s + 4 * (l >> 4)
                m[r, c0:c0 + 4] = halves[t, s, l, 0:2].reshape(4) + halves[t, s, l, 2:4].reshape(4)
    return m


def _split(x):
    h1 = x.astype(np.float16)
    return h1.astype(np.float64), (x - h1.astype(np.float32)).astype(np.float16).astype(np.float64)


def _decode(im, old_tiles, tiles):
    new = im[WP_FL + old_tiles * TILE_FL + 2000 + 480:]           # tower2_kernel's image comes first
    assert new.size >= WP_FL + CS_FL + tiles * TILE_FL + TAPS_FL
    wp = _two_term(new[:WP_FL], 5)
    cs = new[WP_FL:WP_FL + CS_FL].reshape(4, 96)
    wh = _two_term(new[WP_FL + CS_FL:WP_FL + CS_FL + tiles * TILE_FL], tiles) if tiles else None
    taps = new[WP_FL + CS_FL + tiles * TILE_FL:][:5 * 4 * 27 * 4].reshape(5, 4, 27, 4).transpose(2, 0, 1, 3).reshape(27, 80)
    return wp, cs, wh, taps


def _model(x, wp, cs, wh, taps):
    """x (72, H, W) float32 -> BN'd tower-half output (72, H, W) [and output-conv logits] by the kernel's arithmetic"""
    xt = torch.from_numpy(x)[None]
    k = torch.from_numpy(taps[:25, :72].T.reshape(72, 1, 5, 5).copy())
    d = F.conv2d(xt, k, None, 1, 2, 1, 72)[0].numpy()                                   # fp32 depthwise (summation order: tolerance)
    u = np.maximum(d * taps[25, :72, None, None] + taps[26, :72, None, None], 0).astype(np.float32)   # BN x 16, ReLU
    u1, u2 = _split(u.reshape(72, -1))
    w = wp[:72, :72]
    w1 = w.astype(np.float16).astype(np.float64)
    w2 = w - w1
    acc = (w1 @ u2 + w2 @ u1 + w1 @ u1).astype(np.float32)                             # exact products, wide accumulation
    t = acc * cs[0, :72, None] + cs[1, :72, None]
    out = [t.reshape(x.shape)]
    if wh is not None:
        # a half that ends in an output conv: the kernel applies the host-merged matrix (output conv x BN x pointwise conv) to the
        # depthwise result's two fp16 terms directly - `wh` IS that matrix, cs[2] its bias (the 72 x 72 filter is not used)
        h1 = wh[:, :72].astype(np.float16).astype(np.float64)
        h2 = wh[:, :72] - h1
        logit = (h1 @ u2 + h2 @ u1 + h1 @ u1).astype(np.float32) * cs[3, 0] + cs[2, :wh.shape[0], None]
        out.append(logit)
    return out


@pytest.mark.parametrize("classes,tiles", [(80, 6), (5, 1)])
def test_towerh_host_packing_and_arithmetic_vs_oracle(classes, tiles):
    from oracle import yfv2_oracle as oracle
    w = yfv2.random_state_dict(21, classes=classes)
    for k in list(w):                                                       # BatchNorms that actually shift and scale
        if k.startswith("fpn.") and k.endswith("running_mean"):
            w[k] = torch.randn_like(w[k]) * 0.1
        if k.startswith("fpn.") and k.endswith("running_var"):
            w[k] = torch.rand_like(w[k]) + 0.5
    ims = _images(w, classes)
    torch.manual_seed(3)
    # 22x22: the a halves of the two towers side by side in one launch, the b halves in another: the b halves' images are both
    # packed for the wider output conv of the level (`tiles`; tower2_kernel's image in front keeps the half's own)
    for tower, head_keys, mh in (("cls_head_2", ("output_obj_layers", "output_cls_layers"), 3 + classes), ("reg_head_2", ("output_reg_layers",), 12)):
        p = "fpn.%s.block" % tower
        name_a = [n for n in ims if n.startswith(p + " half a")][0]
        name_b = [n for n in ims if n.startswith(p + " half b")][0]
        own = 1 if mh <= 16 else 6
        x = (torch.randn(72, 22, 22) * 2).numpy()
        wp, cs, wh, taps = _decode(ims[name_a], 0, 0)
        got = _model(x, wp, cs, wh, taps)[0]
        xt = torch.from_numpy(x)[None]
        ref = oracle._conv_bn(w, p + ".3", p + ".4", oracle._conv_bn(w, p + ".0", p + ".1", xt, 1, 2, 72, relu=True))
        assert np.abs(got - ref[0].numpy()).max() <= 3e-6 * max(1.0, float(ref.abs().max()))
        # the filter's power of two: largest entry in (2^13, 2^14]
        assert 2.0 ** 13 < np.abs(wp).max() <= 2.0 ** 14
        wp, cs, wh, taps = _decode(ims[name_b], own, tiles)
        t, logit = _model(ref[0].numpy(), wp, cs, wh, taps)
        ref_t = oracle._conv_bn(w, p + ".8", p + ".9", oracle._conv_bn(w, p + ".5", p + ".6", ref, 1, 2, 72, relu=True))
        assert np.abs(t - ref_t[0].numpy()).max() <= 3e-6 * max(1.0, float(ref_t.abs().max()))
        ref_l = torch.cat([F.conv2d(ref_t, w[h + ".weight"], w[h + ".bias"]) for h in head_keys], 1)[0].reshape(mh, -1).numpy()
        assert np.abs(logit[:mh] - ref_l).max() <= 5e-6 * max(1.0, np.abs(ref_l).max())
    # 11x11: the four halves share one launch (first job's image is the step's): every image packed for the level's widest
    # output conv - a half without one carries zero tiles
    name = [n for n in ims if n.startswith("fpn towers 11x11")][0]
    wp, cs, wh, taps = _decode(ims[name], 0, tiles)
    assert wh is not None and not wh.any()
    x = (torch.randn(72, 11, 11) * 2).numpy()
    got = _model(x, wp, cs, None, taps)[0]
    p = "fpn.cls_head_3.block"
    ref = oracle._conv_bn(w, p + ".3", p + ".4", oracle._conv_bn(w, p + ".0", p + ".1", torch.from_numpy(x)[None], 1, 2, 72, relu=True))
    assert np.abs(got - ref[0].numpy()).max() <= 3e-6 * max(1.0, float(ref.abs().max()))
