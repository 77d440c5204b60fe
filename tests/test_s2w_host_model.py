"""CPU check of the fused 96 -> 192 stride-2 launch (block_s2w_kernel, stage4.0): a numpy model that reads ONLY the image
the host packed for it (W1 pre-split into bf16 hi/mid/lo operand quads | W2 | Wproj | main taps | proj taps | ten BN vectors
- yfv2_debug_plan_image) and the channel
order the plan reports for the block's input (yfv2_debug_plan_c2_label: the stage-3 chain leaves C2 permuted), against the
oracle's block.  Pins the packing and the per-input-channel re-ordering; the HIP code itself needs the GPU tests."""
import ctypes as C

import numpy as np
import pytest
import torch

import yolo_fastestv2_amd as yfv2
from oracle import yfv2_oracle as oracle
from yolo_fastestv2_amd import _lib
from yolo_fastestv2_amd._lib import Config, TensorDesc

CIN, KC = 96, 6
W_FL = KC * KC * 256
W1P_FL = KC * (KC // 2) * 3 * 256
IMG_FL = W1P_FL + 2 * W_FL + 2 * 9 * CIN + 10 * CIN


def _plan(w):
    host = {k: v.float().contiguous() for k, v in w.items() if v.is_floating_point()}
    arr = (TensorDesc * len(host))()
    for i, (k, t) in enumerate(host.items()):
        arr[i].name, arr[i].data, arr[i].numel = k.encode(), t.data_ptr(), t.numel()
    cfg = Config()
    cfg.classes, cfg.anchor_num, cfg.height, cfg.width, cfg.max_batch, cfg.device = 80, 3, 352, 352, 1, 0
    L = _lib.lib()
    ns, nb = C.c_int32(0), C.c_int64(0)
    assert L.yfv2_debug_plan_dryrun(C.byref(cfg), arr, len(host), C.byref(ns), C.byref(nb)) == 0
    name = C.create_string_buffer(256)
    buf = np.zeros(IMG_FL, np.float32)
    im = None
    for st in range(ns.value):
        n = L.yfv2_debug_plan_image(C.byref(cfg), arr, len(host), st, name, 256, buf.ctypes.data_as(C.c_void_p), buf.size)
        if n > 0 and "stage4.0 fused s2 block" in name.value.decode():
            assert n == IMG_FL
            im = buf.copy()
            break
    lab = (C.c_int32 * 96)()
    rc = L.yfv2_debug_plan_c2_label(C.byref(cfg), arr, len(host), lab)
    return im, rc, np.asarray(list(lab))


def _frag_matrix(fr):
    """fragment-major [mt][s][lane][4] -> matrix: row 16 mt + (l & 15), column 16 s + 4 (l >> 4) + j"""
    m = np.zeros((CIN, CIN), np.float32)
    fr = fr.reshape(KC, KC, 64, 4)
    for mt in range(KC):
        for s in range(KC):
            for l in range(64):
                m[16 * mt + (l & 15), 16 * s + 4 * (l >> 4):16 * s + 4 * (l >> 4) + 4] = fr[mt, s, l]
    return m


def _presplit_matrix(fr):
    """[mt][chunk pair][term hi, mid, lo][lane][4 dwords]; dword d = two truncated bf16 (low half first) of columns
    16 s + 4 (l >> 4) + 2 (d & 1) + {0, 1}, s = 2 sp + (d >> 1).  Returns (hi + mid + lo, hi, mid, lo) as float32 matrices."""
    u = fr.view(np.uint32).reshape(KC, KC // 2, 3, 64, 4)
    terms = np.zeros((3, CIN, CIN), np.float32)
    for mt in range(KC):
        for sp in range(KC // 2):
            for l in range(64):
                for d in range(4):
                    c = 16 * (2 * sp + (d >> 1)) + 4 * (l >> 4) + 2 * (d & 1)
                    for e in range(2):
                        bits = ((u[mt, sp, :, l, d] >> (16 * e)) & 0xFFFF).astype(np.uint32) << 16
                        terms[:, 16 * mt + (l & 15), c + e] = bits.view(np.float32)
    return (terms[0] + terms[1]) + terms[2], terms


def _dw_s2(x, taps):
    """3x3 stride-2 pad-1 depthwise on (H, W, C) with taps [9][C]"""
    H, W, _ = x.shape
    pad = np.zeros((H + 2, W + 2, x.shape[2]), np.float32)
    pad[1:-1, 1:-1] = x
    d = np.zeros((H // 2, W // 2, x.shape[2]), np.float32)
    for k in range(9):
        d += pad[k // 3:k // 3 + H:2, k % 3:k % 3 + W:2] * taps[k]
    return d


def test_fused_96_channel_stride2_block_host_packing():
    chain = "1"   # the input arrives in the stage-3 chain's channel order (the only plan that has this launch)
    w = yfv2.random_state_dict(11)
    im, rc, lab = _plan(w)
    if im is None:
        pytest.skip("this build's plan has no fused 96-channel stride-2 launch")
    assert rc == (1 if chain == "1" else 0)     # 1 = the plan permutes C2 (the chain), 0 = natural order; lab is filled either way
    assert sorted(lab.tolist()) == list(range(96)) and np.array_equal(lab, np.arange(96)) == (chain == "0")
    torch.manual_seed(3)
    x = torch.randn(1, CIN, 22, 22)
    ref = oracle._shuffle_block(w, "backbone.stage4.0", x, 2)[0].permute(1, 2, 0).numpy()
    xin = x[0].permute(1, 2, 0).numpy()[..., lab]          # position k of the NHWC input holds logical channel lab[k]

    w1, terms = _presplit_matrix(im[:W1P_FL])
    folded = w["backbone.stage4.0.branch_main.0.weight"].reshape(CIN, CIN).numpy()[:, lab]
    assert np.array_equal(w1, folded), "hi + mid + lo must reproduce the fp32 filter exactly"
    assert all(np.array_equal(t.view(np.uint32) & 0xFFFF, np.zeros_like(t, np.uint32)) for t in terms)
    w2, wj = (_frag_matrix(im[W1P_FL + i * W_FL:W1P_FL + (i + 1) * W_FL]) for i in range(2))
    o = W1P_FL + 2 * W_FL
    wd = im[o:o + 9 * CIN].reshape(9, CIN); o += 9 * CIN
    we = im[o:o + 9 * CIN].reshape(9, CIN); o += 9 * CIN
    cs = im[o:].reshape(10, CIN)                           # sc1 sh1 scd shd sc2 sh2 scpd shpd scpp shpp
    proj = _dw_s2(xin, we) * cs[6] + cs[7]
    proj = np.maximum(proj @ wj.T * cs[8] + cs[9], 0.0)
    t1 = np.maximum(xin @ w1.T * cs[0] + cs[1], 0.0)
    main = _dw_s2(t1, wd) * cs[2] + cs[3]
    main = np.maximum(main @ w2.T * cs[4] + cs[5], 0.0)
    got = np.concatenate([proj, main], -1)
    err = np.abs(got - ref).max()
    assert err <= 1e-4 * max(1.0, np.abs(ref).max()), "fused s2 block model vs oracle: max abs err %g" % err


def test_band_rows_fit_the_static_bounds():
    """the kernel's static bounds for the sizes the configuration check admits (input H, W multiples of 32 up to 352):
    a dry run must plan every one of them (the launcher's LDS request and the staging registers are sized from R)."""
    w = yfv2.random_state_dict(11)
    host = {k: v.float().contiguous() for k, v in w.items() if v.is_floating_point()}
    arr = (TensorDesc * len(host))()
    for i, (k, t) in enumerate(host.items()):
        arr[i].name, arr[i].data, arr[i].numel = k.encode(), t.data_ptr(), t.numel()
    L = _lib.lib()
    for H, W in ((352, 352), (320, 320), (288, 384), (64, 96), (32, 32), (352, 32)):
        cfg = Config()
        cfg.classes, cfg.anchor_num, cfg.height, cfg.width, cfg.max_batch, cfg.device = 80, 3, H, W, 1, 0
        ns, nb = C.c_int32(0), C.c_int64(0)
        assert L.yfv2_debug_plan_dryrun(C.byref(cfg), arr, len(host), C.byref(ns), C.byref(nb)) == 0


def _presplit_general(fr, MT, K):
    """[mt][chunk pair][term 2][lane][4 dwords of fp16 pairs] -> (16 MT, K) float64 = first + second fp16 term"""
    KP = K // 32
    u = fr.view(np.uint32).reshape(MT, KP, 2, 64, 4)
    terms = np.zeros((2, 16 * MT, K), np.float64)
    for mt in range(MT):
        for sp in range(KP):
            for l in range(64):
                for d in range(4):
                    c = 16 * (2 * sp + (d >> 1)) + 4 * (l >> 4) + 2 * (d & 1)
                    for e in range(2):
                        bits = ((u[mt, sp, :, l, d] >> (16 * e)) & 0xFFFF).astype(np.uint16)
                        terms[:, 16 * mt + (l & 15), c + e] = bits.view(np.float16).astype(np.float64)
    return terms[0] + terms[1]


@pytest.mark.parametrize("case", ["conv1x1_3", "conv1x1_2 C3 part", "conv1x1_2 C2 part", "conv1x1_2 K=288 (layer by layer)"])
def test_streamed_pointwise_filters_are_packed_presplit(case):
    """pw_kernel<.., PRE>: the FPN reduces' filters arrive as two fp16 terms x 2^sw per chunk pair (fp16x3); their sum must
    reproduce the scaled fp32 filter to 2^-22 of its largest entry (largest entry in (2^13, 2^14]), the BN scale must carry the
    exact 2^-(sw+4), the C2 columns of conv1x1_2 in the chain's channel order.  Default plan (round 6): conv1x1_3 and the C3
    columns of conv1x1_2 are ONE ten-tile launch on C3 (PW_DUAL), the C2 columns a K = 96 launch (PW_FPNQ) whose
    epilogue adds the former; the layer-by-layer plan keeps conv1x1_2 as one K = 288 launch."""
    w = yfv2.random_state_dict(12)
    host = {k: v.float().contiguous() for k, v in w.items() if v.is_floating_point()}
    arr = (TensorDesc * len(host))()
    for i, (k, t) in enumerate(host.items()):
        arr[i].name, arr[i].data, arr[i].numel = k.encode(), t.data_ptr(), t.numel()
    cfg = Config()
    cfg.classes, cfg.anchor_num, cfg.height, cfg.width, cfg.max_batch, cfg.device = 80, 3, 352, 352, 1, 0
    L = _lib.lib()
    plan = _lib.make_plan({"layer_by_layer": 1} if "288" in case else {})
    ns, nb = C.c_int32(0), C.c_int64(0)
    assert L.yfv2_debug_plan_dryrun_ex(C.byref(cfg), C.byref(plan), arr, len(host), C.byref(ns), C.byref(nb)) == 0
    MT = 5
    name, K, half = {"conv1x1_3": ("fpn.conv1x1_3 pw192", 192, 0), "conv1x1_2 C3 part": ("fpn.conv1x1_3 pw192", 192, 1),
                     "conv1x1_2 C2 part": ("fpn.conv1x1_2 pw96", 96, 0), "conv1x1_2 K=288 (layer by layer)": ("fpn.conv1x1_2 up2x", 288, 0)}[case]
    fl = MT * (K // 32) * 2 * 256 + 2 * 16 * MT
    nimg = 2 if K == 192 else 1
    buf = np.zeros(nimg * fl, np.float32)
    nm = C.create_string_buffer(256)
    im = None
    for st in range(ns.value):
        n = L.yfv2_debug_plan_image_ex(C.byref(cfg), C.byref(plan), arr, len(host), st, nm, 256, buf.ctypes.data_as(C.c_void_p), buf.size)
        if n > 0 and name in nm.value.decode():
            assert n == nimg * fl, (n, nimg * fl)
            if nimg == 2:   # PW_DUAL: fragments of the first conv, of the second, then scale[2][80], shift[2][80]
                fr, r = fl - 2 * 16 * MT, 16 * MT
                im = np.concatenate([buf[half * fr:(half + 1) * fr], buf[2 * fr + half * r:2 * fr + (half + 1) * r],
                                     buf[2 * fr + 2 * r + half * r:2 * fr + 2 * r + (half + 1) * r]])
            else:
                im = buf.copy()
            break
    assert im is not None
    got = _presplit_general(im[:fl - 2 * 16 * MT], MT, K)[:72]
    w2 = w["fpn.conv1x1_2.0.weight"].reshape(72, 288).numpy()
    lab = (C.c_int32 * 96)()
    assert L.yfv2_debug_plan_c2_label(C.byref(cfg), arr, len(host), lab) == 1
    w2c2 = w2[:, 192:][:, np.asarray(list(lab))]                    # cat(up(C3), C2): C2 columns in the chain's channel order
    if "288" in case:
        w2c2 = w2[:, 192:]                                           # (layer by layer there is no chain kernel: the reference's own order)
    ref = {"conv1x1_3": w["fpn.conv1x1_3.0.weight"].reshape(72, 192).numpy(), "conv1x1_2 C3 part": w2[:, :192], "conv1x1_2 C2 part": w2c2,
           "conv1x1_2 K=288 (layer by layer)": np.concatenate([w2[:, :192], w2c2], 1)}[case]
    sw = 14 - int(np.ceil(np.log2(np.abs(ref).max())))
    assert 2.0 ** 13 < np.abs(got).max() <= 2.0 ** 14
    assert np.abs(got - ref.astype(np.float64) * 2.0 ** sw).max() <= 2.0 ** 14 * 2.0 ** -22
    bn = "fpn.conv1x1_3.1" if case == "conv1x1_3" else "fpn.conv1x1_2.1"
    scale = (w[bn + ".weight"] / torch.sqrt(w[bn + ".running_var"] + 1e-5)).numpy()
    assert np.allclose(im[fl - 2 * 16 * MT:fl - 16 * MT][:72] * 2.0 ** (sw + 4), scale, rtol=1e-6, atol=0)
    shift = w[bn + ".bias"].numpy() - w[bn + ".running_mean"].numpy() * scale
    assert np.allclose(im[fl - 16 * MT:][:72], shift, rtol=1e-5, atol=1e-7)
