"""CPU check of the stage-4 chain launch (block_s1pool_kernel): a numpy model that reads ONLY what the host packed for it
(per block three 32-channel passes: W1 rows, W2 columns, taps, BN vectors - yfv2_debug_plan_image) and runs the kernel's
pass structure (pw1 of a pass -> tile -> depthwise -> pw2 partial sums; pool <- cat(even channels, fresh)) against the
oracle's three stride-1 blocks.  Pins the per-pass slicing / fragment packing; the HIP code itself needs the GPU tests."""
import ctypes as C

import numpy as np
import pytest
import torch

import yolo_fastestv2_amd as yfv2
from oracle import yfv2_oracle as oracle
from yolo_fastestv2_amd import _lib
from yolo_fastestv2_amd._lib import Config, TensorDesc

C2, NB = 96, 3
REST_FL = 9 * 32 + 4 * 32 + 2 * C2
# fp32 fragments (YFV2_BF6=0) / two fp16 terms x 2^sw per chunk pair (default, fp16x3): [mt][pair][term][64][4]
SIZES = {False: (2 * 6 * 256, 6 * 2 * 256), True: (2 * 3 * 2 * 256, 6 * 1 * 2 * 256)}


def _plan_image(w):
    host = {k: v.float().contiguous() for k, v in w.items() if v.is_floating_point()}
    arr = (TensorDesc * len(host))()
    for i, (k, t) in enumerate(host.items()):
        arr[i].name, arr[i].data, arr[i].numel = k.encode(), t.data_ptr(), t.numel()
    cfg = Config()
    cfg.classes, cfg.anchor_num, cfg.height, cfg.width, cfg.max_batch, cfg.device = 80, 3, 352, 352, 1, 0
    L = _lib.lib()
    ns, nb = C.c_int32(0), C.c_int64(0)
    plan = _lib.make_plan(_lib.plan_from_env())     # (the library reads no environment: the Python layer maps YFV2_BF6=0 onto yfv2_plan.fp32_matrix)
    assert L.yfv2_debug_plan_dryrun_ex(C.byref(cfg), C.byref(plan), arr, len(host), C.byref(ns), C.byref(nb)) == 0
    name = C.create_string_buffer(256)
    buf = np.zeros(NB * 3 * (sum(SIZES[True]) + REST_FL), np.float32)
    for st in range(ns.value):
        n = L.yfv2_debug_plan_image_ex(C.byref(cfg), C.byref(plan), arr, len(host), st, name, 256, buf.ctypes.data_as(C.c_void_p), buf.size)
        if n > 0 and "whole activation resident in LDS" in name.value.decode():
            return buf[:n].copy()          # (n = what fitted of the blob from the image's start on, not the image's length)
    return None


def _presplit(fr, mt_n, kp):
    """[mt][chunk pair][term 2][lane][4 dwords] -> (16 mt_n, 32 kp) float64 = the sum of the two fp16 terms (the filter x 2^sw);
    dword d = two fp16 (low half first) of columns 16 (2 sp + (d >> 1)) + 4 (l >> 4) + 2 (d & 1) + {0, 1}"""
    u = fr.view(np.uint32).reshape(mt_n, kp, 2, 64, 4)
    terms = np.zeros((2, 16 * mt_n, 32 * kp))
    for mt in range(mt_n):
        for sp in range(kp):
            for l in range(64):
                for d in range(4):
                    c = 16 * (2 * sp + (d >> 1)) + 4 * (l >> 4) + 2 * (d & 1)
                    for e in range(2):
                        bits = ((u[mt, sp, :, l, d] >> (16 * e)) & 0xFFFF).astype(np.uint16)
                        terms[:, 16 * mt + (l & 15), c + e] = bits.view(np.float16).astype(np.float64)
    return terms[0] + terms[1]


def _frags(fr, mt_n, s_n):
    """fragment-major [mt][s][lane][4] -> (16 mt_n, 16 s_n) matrix: row 16 mt + (l & 15), column 16 s + 4 (l >> 4) + j"""
    fr = fr.reshape(mt_n, s_n, 64, 4)
    m = np.zeros((16 * mt_n, 16 * s_n), np.float32)
    for mt in range(mt_n):
        for s in range(s_n):
            for l in range(64):
                m[16 * mt + (l & 15), 16 * s + 4 * (l >> 4):16 * s + 4 * (l >> 4) + 4] = fr[mt, s, l]
    return m


@pytest.mark.parametrize("form", ["presplit", "fp32"])
def test_pool_chain_host_packing(monkeypatch, form):
    if form == "fp32":
        monkeypatch.setenv("YFV2_BF6", "0")
    w = yfv2.random_state_dict(9)
    im = _plan_image(w)
    if im is None:
        pytest.skip("this build's plan has no stage-4 chain launch")
    pre = form == "presplit"            # the default plan packs pre-split; YFV2_BF6=0 the fp32 fragments
    W1_FL, W2_FL = SIZES[pre]
    IMG_FL = W1_FL + W2_FL + REST_FL
    torch.manual_seed(2)
    x = torch.randn(1, 192, 11, 11)
    ref = x
    for k in range(1, 4):
        ref = oracle._shuffle_block(w, "backbone.stage4.%d" % k, ref, 1)
    pool = x[0].permute(1, 2, 0).contiguous().numpy()
    H, W, _ = pool.shape
    for blk in range(NB):
        bi = pool[..., 1::2]                                     # the odd channels: the branch input
        acc2 = np.zeros((H, W, C2), np.float32)
        for th in range(3):
            t = im[(blk * 3 + th) * IMG_FL:(blk * 3 + th + 1) * IMG_FL]
            w1t = _presplit(t[:W1_FL], 2, 3) if pre else _frags(t[:W1_FL], 2, 6)                        # (32, 96): rows 32 th .. +31
            w2t = _presplit(t[W1_FL:W1_FL + W2_FL], 6, 1) if pre else _frags(t[W1_FL:W1_FL + W2_FL], 6, 2)  # (96, 32): columns 32 th .. +31
            up = 16.0 if pre else 1.0   # fp16x3: activations x 2^4, filters x 2^sw, both undone inside the BN scales
            if pre:   # the two terms reproduce the filter x 2^sw (largest entry in (2^13, 2^14]) to 2^-22 of that
                p4 = "backbone.stage4.%d.branch_main." % (blk + 1)
                for got, full, sl in ((w1t, w[p4 + "0.weight"].reshape(C2, C2).numpy(), np.s_[32 * th:32 * th + 32, :]),
                                      (w2t, w[p4 + "5.weight"].reshape(C2, C2).numpy(), np.s_[:, 32 * th:32 * th + 32])):
                    sw = 14 - int(np.ceil(np.log2(np.abs(full).max())))
                    assert np.abs(got - full[sl].astype(np.float64) * 2.0 ** sw).max() <= 2.0 ** 14 * 2.0 ** -22
            o = W1_FL + W2_FL
            taps = t[o:o + 288].reshape(9, 32)
            sc1, sh1, scd, shd = t[o + 288:o + 416].reshape(4, 32)
            sc2, sh2 = t[o + 416:o + 416 + 2 * C2].reshape(2, C2)
            y = np.maximum((bi * up) @ w1t.T * sc1 + sh1, 0.0).astype(np.float32)
            pad = np.zeros((H + 2, W + 2, 32), np.float32)
            pad[1:-1, 1:-1] = y
            d = np.zeros((H, W, 32), np.float32)
            for k in range(9):
                d += pad[k // 3:k // 3 + H, k % 3:k % 3 + W] * taps[k]
            acc2 += ((d * scd + shd) * up) @ w2t.T
        fresh = np.maximum(acc2 * sc2 + sh2, 0.0).astype(np.float32)
        pool = np.concatenate((pool[..., 0::2], fresh), -1)
    want = ref[0].permute(1, 2, 0).numpy()
    err = np.abs(pool - want).max()
    assert err <= 1e-4 * max(1.0, np.abs(want).max()), "pool-chain model vs oracle: max abs err %g" % err
