"""s1h_kernel (yfv2_stage2h.hip: stage 2's stride-1 blocks with both pointwise convs on the f16 matrix cores) pinned on the
CPU: the image the host packs for it (yfv2_debug_plan_image; it follows s1px_kernel's image in the blob) is decoded - the
two-term fp16 filters in MFMA A-operand order, the quad-packed depthwise taps, the scaled BN shifts, the per-lane pair
offsets - and a numpy model of the KERNEL's dataflow in its own channel-position space (lane group g owns positions
4g..4g+3 and 16+4g..16+4g+3; powers of two as the kernel applies them) must reproduce the oracle's block."""
import ctypes as C

import numpy as np
import torch
import torch.nn.functional as F

import yolo_fastestv2_amd as yfv2
from oracle import yfv2_oracle as oracle
from yolo_fastestv2_amd import _lib
from yolo_fastestv2_amd._lib import Config, TensorDesc

OLD_FL = 1280 + 54 * 64
NEW_FL = 3272 + 512


def npos(g, j):
    return 4 * g + j if j < 4 else (16 + 4 * g + (j - 4) if g < 2 else -1)


def _plan_image(w, block):
    host = {k: v.float().contiguous() for k, v in w.items() if v.is_floating_point()}
    arr = (TensorDesc * len(host))()
    for i, (k, t) in enumerate(host.items()):
        arr[i].name, arr[i].data, arr[i].numel = k.encode(), t.data_ptr(), t.numel()
    cfg = Config()
    cfg.classes, cfg.anchor_num, cfg.height, cfg.width, cfg.max_batch, cfg.device = 80, 3, 352, 352, 1, 0
    cap = OLD_FL + NEW_FL
    buf = np.zeros(cap, np.float32)
    name = C.create_string_buffer(256)
    for step in range(1, 6):
        n = _lib.lib().yfv2_debug_plan_image(C.byref(cfg), arr, len(host), step, name, 256, buf.ctypes.data_as(C.c_void_p), cap)
        if name.value.decode().startswith("backbone.stage2.%d s1 block" % block):
            assert n == cap
            return buf[OLD_FL:].copy()
    raise AssertionError("no lane-per-pixel stage2.%d launch in the plan" % block)


def _decode_filter(fl):
    u = fl.view(np.uint32).reshape(2, 2, 64, 4)
    halves = np.stack((u & 0xffff, u >> 16), -1).astype(np.uint16).view(np.float16).astype(np.float64)   # [t][term][lane][d][e]
    m = np.zeros((32, 24))
    for t in range(2):
        for l in range(64):
            for j in range(8):
                n = npos(l >> 4, j)
                v = halves[t, 0, l, j // 2, j % 2] + halves[t, 1, l, j // 2, j % 2]
                if n >= 0:
                    m[16 * t + (l & 15), n] = v
                else:
                    assert v == 0
    return m


def test_s1h_host_packing_and_dataflow_vs_oracle():
    w = yfv2.random_state_dict(6)
    for k in list(w):
        if k.startswith("backbone.stage2.") and k.endswith("running_mean"):
            w[k] = torch.randn_like(w[k]) * 0.1
            w[k.replace("running_mean", "running_var")] = torch.rand_like(w[k]) + 0.5
            w[k.replace("running_mean", "weight")] = torch.randn_like(w[k])
            w[k.replace("running_mean", "bias")] = torch.randn_like(w[k]) * 0.2
    for block in (1, 2, 3):
        im = _plan_image(w, block)
        W1, W2 = _decode_filter(im[0:1024]), _decode_filter(im[1024:2048])
        taps = np.zeros((4, 8, 9))
        for q in range(18):
            for l in range(64):
                f = 4 * q + (l & 3)
                taps[l >> 4, f // 9, f % 9] = im[2048 + q * 64 + l]
        for l in range(64):   # all quads of a lane group carry the same four taps
            assert im[2048 + np.arange(18) * 64 + l].tolist() == im[2048 + np.arange(18) * 64 + (l & 0x33)].tolist()
        sh1, bias2, unscale2 = im[3200:3232].astype(np.float64), im[3232:3264].astype(np.float64), float(im[3264])
        offs = im[3272:].view(np.int32).reshape(2, 4, 64)
        for l in range(64):
            for k in range(4):
                valid = k < 2 or (l >> 4) < 2
                assert (offs[0, k, l] != -2 ** 31) == valid and (offs[1, k, l] != -2 ** 31) == valid
        live = offs[0][offs[0] != -2 ** 31]
        assert len(set(live.tolist())) == 12 and len(set(offs[1][offs[1] != -2 ** 31].tolist())) == 12     # 12 pairs read, 12 written
        assert not set(live.tolist()) & set(offs[1][offs[1] != -2 ** 31].tolist())                         # never in place

        p = "backbone.stage2.%d" % block
        f1 = oracle._conv_bn   # noqa: F841  (the oracle's own fold, for reference)
        # folded reference filters
        def fold(conv, bn):
            g, b, m, v = (w[bn + s].double() for s in (".weight", ".bias", ".running_mean", ".running_var"))
            sc = g / torch.sqrt(v + 1e-5)
            return w[conv + ".weight"].double(), sc, b - m * sc
        w1, sc1, shv1 = fold(p + ".branch_main.0", p + ".branch_main.1")
        w1 = (w1.reshape(24, 24) * sc1[:, None]).numpy()
        # recover sw1 and the position -> branch-input-channel order from W1's columns
        sw1 = int(round(np.log2(np.abs(W1[:24]).max() / np.abs(w1).max())))
        order = []
        for n in range(24):
            err = np.abs(W1[:24, n][:, None] / 2.0 ** sw1 - w1).max(0)
            order.append(int(err.argmin()))
            assert err.min() <= 2e-7 * np.abs(w1).max(), (block, n, err.min())
        assert sorted(order) == list(range(24))
        assert np.allclose(sh1[:24] / 2.0 ** (sw1 + 4), shv1.numpy(), rtol=1e-6, atol=1e-7)

        torch.manual_seed(block)
        x = torch.randn(1, 48, 12, 16).double()
        ref = oracle._shuffle_block({k: v.double() for k, v in w.items() if v.is_floating_point()}, p, x, 1)[0, 24:].numpy()   # the branch's 24 outputs
        xin = x[0, 1::2].numpy()[order]                     # branch input at position n = odd channel order[n]
        # the kernel's arithmetic in position space (exact products: float64 stands in for the two-term fp16 operands)
        t = np.maximum(np.einsum("rn,nyx->ryx", W1[:24], xin * 16.0) + sh1[:24, None, None], 0.0)           # relu(pw1) * 2^(sw1+4)
        tp = np.pad(t, ((0, 0), (1, 1), (1, 1)))
        d = np.zeros_like(t)
        for n in range(24):
            g, cs = (n // 4, n % 4) if n < 16 else ((n - 16) // 4, 4 + (n - 16) % 4)
            for dy in range(3):
                for dx in range(3):
                    d[n] += taps[g, cs, dy * 3 + dx] * tp[n, dy:dy + 12, dx:dx + 16]
        out = np.maximum(np.einsum("rn,nyx->ryx", W2[:24], d) + bias2[:24, None, None], 0.0) * unscale2
        want = ref[order]                                     # output position n = branch output channel order[n]
        err = np.abs(out - want).max()
        assert err <= 2e-6 * max(1.0, np.abs(want).max()), "block %d: dataflow model vs oracle: %g" % (block, err)


# ---------------------------------------------------------------------------------------------------------------------
# s2h_kernel: stage2.0 (24 -> 48, stride 2), both branches in one wave
# ---------------------------------------------------------------------------------------------------------------------
S2_OLD_FL = (640 + 54 * 64) + (1280 + 54 * 64)      # image_s2px_proj + image_s2px_main precede it in the blob
S2_NEW_FL = 5480 + 12 * 64


def _stage2_channel(slot):                           # yfv2_stage2_channel (yfv2_internal.h)
    p, e = slot >> 1, slot & 1
    return p // 12 + 2 * ((p % 12) // 6) + 4 * ((p % 6) // 3) + 8 * (2 * (p % 3) + e)


def _fold(w, conv, bn):
    g, b, m, v = (w[bn + s].double() for s in (".weight", ".bias", ".running_mean", ".running_var"))
    sc = g / torch.sqrt(v + 1e-5)
    return w[conv + ".weight"].double(), sc, b - m * sc


def _match_rows(dec, ref):
    """dec[r] = 2^sw * ref[pos[r]] -> (sw, pos)"""
    sw = int(round(np.log2(np.abs(dec[:24]).max() / np.abs(ref).max())))
    pos = []
    for r in range(24):
        err = np.abs(dec[r][None, :] / 2.0 ** sw - ref).max(1)
        pos.append(int(err.argmin()))
        assert err.min() <= 2e-7 * np.abs(ref).max(), (r, err.min())
    assert sorted(pos) == list(range(24))
    return sw, pos


def test_s2h_host_packing_dataflow_and_slot_map_vs_oracle():
    w = yfv2.random_state_dict(8)
    for k in list(w):
        if k.startswith("backbone.stage2.0") and k.endswith("running_mean"):
            w[k] = torch.randn_like(w[k]) * 0.1
            w[k.replace("running_mean", "running_var")] = torch.rand_like(w[k]) + 0.5
            w[k.replace("running_mean", "weight")] = torch.randn_like(w[k])
            w[k.replace("running_mean", "bias")] = torch.randn_like(w[k]) * 0.2
    host = {k: v.float().contiguous() for k, v in w.items() if v.is_floating_point()}
    arr = (TensorDesc * len(host))()
    for i, (k, t) in enumerate(host.items()):
        arr[i].name, arr[i].data, arr[i].numel = k.encode(), t.data_ptr(), t.numel()
    cfg = Config()
    cfg.classes, cfg.anchor_num, cfg.height, cfg.width, cfg.max_batch, cfg.device = 80, 3, 352, 352, 1, 0
    cap = S2_OLD_FL + S2_NEW_FL
    buf = np.zeros(cap, np.float32)
    name = C.create_string_buffer(256)
    n = _lib.lib().yfv2_debug_plan_image(C.byref(cfg), arr, len(host), 1, name, 256, buf.ctypes.data_as(C.c_void_p), cap)
    assert n == cap and name.value.decode().startswith("backbone.stage2.0 s2 block"), (n, name.value)
    im = buf[S2_OLD_FL:]
    W1, WP, W2 = (_decode_filter(im[o:o + 1024]) for o in (0, 1024, 2048))

    def taps_at(o):
        t = np.zeros((4, 8, 9))
        for q in range(18):
            for l in range(64):
                f = 4 * q + (l & 3)
                t[l >> 4, f // 9, f % 9] = im[o + q * 64 + l]
        return t
    tm, tp = taps_at(3072), taps_at(4224)
    sh1, bip, bi2 = (im[5376 + 32 * i:5376 + 32 * i + 32].astype(np.float64) for i in range(3))
    un_p, un_2 = float(im[5376 + 96]), float(im[5376 + 97])
    offs = im[5480:].view(np.int32).reshape(12, 64)

    p = "backbone.stage2.0"
    w1, sc1, b1 = _fold(w, p + ".branch_main.0", p + ".branch_main.1")
    w2, sc2, b2 = _fold(w, p + ".branch_main.5", p + ".branch_main.6")
    wq, scq, bq = _fold(w, p + ".branch_proj.2", p + ".branch_proj.3")
    sw1, pos_in = _match_rows(W1, (w1.reshape(24, 24) * sc1[:, None]).numpy())
    assert pos_in == list(range(24))                    # pw1's rows and all input positions are natural channels
    swp, pos0 = _match_rows(WP, (wq.reshape(24, 24) * scq[:, None]).numpy())
    sw2, pos1 = _match_rows(W2, (w2.reshape(24, 24) * sc2[:, None]).numpy())

    # slot map: logical output channel c (proj 0..23, main 24..47) lives in pair slot_of[c] >> 1, element slot_of[c] & 1
    slot_of = {_stage2_channel(s): s for s in range(48)}
    OH = OW = 44
    for l in range(64):
        g = l >> 4
        for k in range(4):                              # loads: natural pair planes of the 88x88 input
            kk = 2 * g + k if k < 2 else (8 + 2 * g + (k - 2) if g < 2 else -1)
            assert offs[k, l] == (kk * 88 * 88 * 8 if kk >= 0 else -2 ** 31)
        for k in range(2):                              # whole pairs of either branch: positions 4g + 2k, 4g + 2k + 1
            for role, posr, base in ((0, pos0, 0), (1, pos1, 24)):
                c0, c1 = base + posr[4 * g + 2 * k], base + posr[4 * g + 2 * k + 1]
                assert slot_of[c0] % 2 == 0 and slot_of[c1] == slot_of[c0] + 1
                assert offs[4 + 2 * role + k, l] == (slot_of[c0] >> 1) * OH * OW * 8
        for e in range(4):                              # mixed pairs: position 16 + 4g + e of both branches
            if g < 2:
                cp, cm = pos0[16 + 4 * g + e], 24 + pos1[16 + 4 * g + e]
                assert slot_of[cp] % 2 == 0 and slot_of[cm] == slot_of[cp] + 1
                assert offs[8 + e, l] == (slot_of[cp] >> 1) * OH * OW * 8
            else:
                assert offs[8 + e, l] == -2 ** 31

    # dataflow in position space vs the oracle's block
    torch.manual_seed(1)
    x = torch.randn(1, 24, 16, 24).double()
    wd = {k: v.double() for k, v in w.items() if v.is_floating_point()}
    ref = oracle._shuffle_block(wd, p, x, 2)[0].numpy()                         # (48, 8, 12)
    x16 = x[0].numpy() * 16.0
    t = np.maximum(np.einsum("rn,nyx->ryx", W1[:24], x16) + sh1[:24, None, None], 0.0)

    def dw_s2(v, taps):
        vp = np.pad(v, ((0, 0), (1, 1), (1, 1)))
        d = np.zeros((24, 8, 12))
        for nn in range(24):
            g, cs = (nn // 4, nn % 4) if nn < 16 else ((nn - 16) // 4, 4 + (nn - 16) % 4)
            for dy in range(3):
                for dx in range(3):
                    d[nn] += taps[g, cs, dy * 3 + dx] * vp[nn, dy:dy + 16:2, dx:dx + 24:2]
        return d
    proj = np.maximum(np.einsum("rn,nyx->ryx", WP[:24], dw_s2(x16, tp)) + bip[:24, None, None], 0.0) * un_p
    main = np.maximum(np.einsum("rn,nyx->ryx", W2[:24], dw_s2(t, tm)) + bi2[:24, None, None], 0.0) * un_2
    for got, want in ((proj, ref[:24][pos0]), (main, ref[24:][pos1])):
        err = np.abs(got - want).max()
        assert err <= 2e-6 * max(1.0, np.abs(want).max()), err
