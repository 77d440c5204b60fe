"""CPU check of the two-blocks-in-one-launch step (block_s1x2_kernel): a numpy model of the KERNEL'S dataflow - the
lane-local channel splits, the physical channel order of its LDS tile, the three places it stores to - driven by the
image the HOST packed for that launch (yfv2_debug_plan_image), against the oracle's two stride-1 blocks.  What this pins
is the index algebra shared by kernel and host (yfv2_s1x2_label_a/b, the Z store offsets, the fragment-major filter
packing with permuted input columns); the HIP code itself needs the GPU tests."""
import ctypes as C
import ctypes as C_

import numpy as np
import pytest
import torch

import yolo_fastestv2_amd as yfv2
from oracle import yfv2_oracle as oracle
from yolo_fastestv2_amd import _lib
from yolo_fastestv2_amd._lib import Config, TensorDesc

KC, C2 = 3, 48
W_FL, DW_FL, CST_FL = KC * KC * 256, 9 * KC * 16, 6 * KC * 16
IMG_FL = 2 * W_FL + DW_FL + CST_FL


def _plan_image(w, step_substr):
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
    buf = np.zeros(2 * IMG_FL, np.float32)
    for st in range(ns.value):
        n = L.yfv2_debug_plan_image(C.byref(cfg), arr, len(host), st, name, 256, buf.ctypes.data_as(C.c_void_p), buf.size)
        assert n > 0
        if step_substr in name.value.decode():
            return buf.copy(), name.value.decode()
    return None, None


def _frag_matrix(fr):
    """fragment-major [mt][s][lane][4] -> the matrix the MFMAs see: rows = output channel, columns = PHYSICAL input position 16 s + 4 g + j"""
    m = np.zeros((C2, C2), np.float32)
    fr = fr.reshape(KC, KC, 64, 4)
    for mt in range(KC):
        for s in range(KC):
            for l in range(64):
                for j in range(4):
                    m[16 * mt + (l & 15), 16 * s + 4 * (l >> 4) + j] = fr[mt, s, l, j]
    return m


def _split_image(im):
    w1, w2 = _frag_matrix(im[:W_FL]), _frag_matrix(im[W_FL:2 * W_FL])
    wd = im[2 * W_FL:2 * W_FL + DW_FL].reshape(9, C2)
    cs = im[2 * W_FL + DW_FL:IMG_FL].reshape(6, C2)
    return w1, w2, wd, cs


def _branch(tile_phys, im):
    """tile_phys: (H, W, 48) branch input in PHYSICAL channel order -> (H, W, 48) branch output in logical order"""
    w1, w2, wd, cs = _split_image(im)
    H, W, _ = tile_phys.shape
    y = np.maximum(tile_phys @ w1.T * cs[0] + cs[1], 0.0).astype(np.float32)      # pw1 + BN + ReLU, in place in the tile
    pad = np.zeros((H + 2, W + 2, C2), np.float32)
    pad[1:-1, 1:-1] = y
    d = np.zeros((H, W, C2), np.float32)
    for k in range(9):
        d += pad[k // 3:k // 3 + H, k % 3:k % 3 + W] * wd[k]
    d = d * cs[2] + cs[3]
    return np.maximum(d @ w2.T * cs[4] + cs[5], 0.0).astype(np.float32)


def _kernel_model(x, im):
    """x: (H, W, 96) -> z: (H, W, 96), following block_s1x2_kernel's data movement lane by lane"""
    H, W, _ = x.shape
    z = np.full((H, W, 96), np.nan, np.float32)
    tile = np.zeros((H, W, C2), np.float32)            # physical position 16 s + 4 g + j  (plane 4 s + g, element j)
    hold = np.zeros((H, W, 6, 4), np.float32)
    xq = x.reshape(H, W, 6, 4, 4)                      # [chunk c][lane group g][element]
    for g in range(4):
        for c in range(6):
            z[..., 4 * c + g] = xq[:, :, c, g, 0]
            hold[:, :, c, g] = xq[:, :, c, g, 2]
        for j in range(3):                             # quad j of lane group g -> plane 4 j + g
            tile[..., 16 * j + 4 * g + 0] = xq[:, :, 2 * j, g, 1]
            tile[..., 16 * j + 4 * g + 1] = xq[:, :, 2 * j, g, 3]
            tile[..., 16 * j + 4 * g + 2] = xq[:, :, 2 * j + 1, g, 1]
            tile[..., 16 * j + 4 * g + 3] = xq[:, :, 2 * j + 1, g, 3]
    bo = _branch(tile, im[:IMG_FL]).reshape(H, W, 3, 4, 4)           # [mt][g][r]
    tile_b = np.zeros((H, W, C2), np.float32)
    for g in range(4):
        q0 = [hold[:, :, 0, g], hold[:, :, 1, g], hold[:, :, 2, g], hold[:, :, 3, g]]
        q1 = [hold[:, :, 4, g], hold[:, :, 5, g], bo[:, :, 0, g, 1], bo[:, :, 0, g, 3]]
        q2 = [bo[:, :, 1, g, 1], bo[:, :, 1, g, 3], bo[:, :, 2, g, 1], bo[:, :, 2, g, 3]]
        for s, q in enumerate((q0, q1, q2)):
            for j in range(4):
                tile_b[..., 16 * s + 4 * g + j] = q[j]
        for mt in range(3):
            z[..., 24 + 8 * mt + 2 * g] = bo[:, :, mt, g, 0]
            z[..., 24 + 8 * mt + 2 * g + 1] = bo[:, :, mt, g, 2]
    z[..., 48:] = _branch(tile_b, im[IMG_FL:2 * IMG_FL])
    assert not np.isnan(z).any(), "a Z channel was never stored"
    return z


def test_two_block_step_host_packing_and_index_algebra(monkeypatch):
    monkeypatch.setenv("YFV2_S1CHAIN", "0")   # the plan without the seven-block chain: pairs of blocks
    w = yfv2.random_state_dict(5)
    im, name = _plan_image(w, "stage3.1 + 2 two fused s1 blocks")
    if im is None:
        pytest.skip("this build's plan has no two-block launch")
    torch.manual_seed(0)
    x = torch.randn(1, 96, 22, 22)
    ref = oracle._shuffle_block(w, "backbone.stage3.2", oracle._shuffle_block(w, "backbone.stage3.1", x, 1), 1)
    got = _kernel_model(x[0].permute(1, 2, 0).contiguous().numpy(), im)
    ref = ref[0].permute(1, 2, 0).numpy()
    err = np.abs(got - ref).max()
    assert err <= 1e-4 * max(1.0, np.abs(ref).max()), "two-block dataflow model vs oracle: max abs err %g" % err


def test_fused_depthwise_pointwise_step_host_packing(monkeypatch):
    monkeypatch.setenv("YFV2_S1CHAIN", "0")   # plain NHWC input to the 96 -> 192 block (the chain permutes C2, tests/test_s1chain_host_model.py)
    monkeypatch.setenv("YFV2_S2W", "0")       # the plan with stage4.0 as three launches (the default fuses it: tests/test_s2w_host_model.py)
    """dwpw_s2_kernel's image (pointwise fragments | depthwise taps [9][C] | dw scale, shift | pw scale, shift) for the two
    branch tails of the 96 -> 192 block: a numpy model of dw3x3 s2 + BN -> pw + BN + ReLU reading that image, against the
    oracle's layers."""
    w = yfv2.random_state_dict(6)
    C = 96
    frag_fl = (C // 16) ** 2 * 256
    torch.manual_seed(1)
    x = torch.randn(1, C, 22, 22)
    for substr, pre, convs in (("stage4.0.proj: dw3x3s2", None, ("branch_proj.0", "branch_proj.1", "branch_proj.2", "branch_proj.3")),
                               ("stage4.0.main: dw3x3s2", None, ("branch_main.3", "branch_main.4", "branch_main.5", "branch_main.6"))):
        host = {k: v.float().contiguous() for k, v in w.items() if v.is_floating_point()}
        arr = (TensorDesc * len(host))()
        for i, (k, t) in enumerate(host.items()):
            arr[i].name, arr[i].data, arr[i].numel = k.encode(), t.data_ptr(), t.numel()
        cfg = Config()
        cfg.classes, cfg.anchor_num, cfg.height, cfg.width, cfg.max_batch, cfg.device = 80, 3, 352, 352, 1, 0
        L = _lib.lib()
        ns, nb = C_.c_int32(0), C_.c_int64(0)
        assert L.yfv2_debug_plan_dryrun(C_.byref(cfg), arr, len(host), C_.byref(ns), C_.byref(nb)) == 0
        name = C_.create_string_buffer(256)
        buf = np.zeros(frag_fl + 13 * C, np.float32)
        im = None
        for st in range(ns.value):
            n = L.yfv2_debug_plan_image(C_.byref(cfg), arr, len(host), st, name, 256, buf.ctypes.data_as(C_.c_void_p), buf.size)
            if n > 0 and substr in name.value.decode():
                im = buf.copy()
                break
        if im is None:
            pytest.skip("this build's plan has no fused depthwise+pointwise launch")
        KCc = C // 16
        fr = im[:frag_fl].reshape(KCc, KCc, 64, 4)
        wp = np.zeros((C, C), np.float32)
        for mt in range(KCc):
            for s in range(KCc):
                for l in range(64):
                    wp[16 * mt + (l & 15), 16 * s + 4 * (l >> 4):16 * s + 4 * (l >> 4) + 4] = fr[mt, s, l]
        taps = im[frag_fl:frag_fl + 9 * C].reshape(9, C)
        cs = im[frag_fl + 9 * C:].reshape(4, C)
        xin = x[0].permute(1, 2, 0).numpy()
        pad = np.zeros((24, 24, C), np.float32)
        pad[1:-1, 1:-1] = xin
        d = np.zeros((11, 11, C), np.float32)
        for k in range(9):
            d += pad[k // 3:k // 3 + 22:2, k % 3:k % 3 + 22:2] * taps[k]
        d = d * cs[0] + cs[1]
        got = np.maximum(d @ wp.T * cs[2] + cs[3], 0.0)
        p = "backbone.stage4.0."
        ref = oracle._conv_bn(w, p + convs[0], p + convs[1], x, 2, 1, C)
        ref = oracle._conv_bn(w, p + convs[2], p + convs[3], ref, relu=True)[0].permute(1, 2, 0).numpy()
        err = np.abs(got - ref).max()
        assert err <= 1e-4 * max(1.0, np.abs(ref).max()), "%s: max abs err %g" % (substr, err)
