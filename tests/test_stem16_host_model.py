"""stem_h3_kernel (yfv2_stem16.hip: the stem as an implicit GEMM on the f16 matrix cores, every operand split into two fp16
terms) pinned on the CPU: a numpy model of the KERNEL's dataflow - the lane's three 16-byte loads per conv row, the eight K
slots per lane group built from them (lane group 3 carrying tap (2,2) of the three channels), three exact f16 x f16 products per MAC accumulated in fp32, BN shift as the initial accumulator, max-pool before
ReLU, the power-of-two unscale - driven ONLY by the image the host packed (yfv2_debug_plan_image), against the oracle's
stem.  Also the arithmetic claim of the kernel's header: fp16x3 is as close to float64 as the fp32 convolution is."""
import ctypes as C

import numpy as np
import torch
import torch.nn.functional as F

import yolo_fastestv2_amd as yfv2
from oracle import yfv2_oracle as oracle
from yolo_fastestv2_amd import _lib
from yolo_fastestv2_amd._lib import Config, TensorDesc

OLD_IMG_FL = 11 * 64 + 24                  # image_stem (4x4x1 kernel), packed twice (fp32 and uint8 scale) before the fp16 image
H16_FL = 2 * 2 * 64 * 4 + 36 + 36          # filter terms | fp32-input constants (32 shifts, unscale, pad) | uint8-input constants
TAP = [(0, 1), (0, 2), (1, 1), (1, 2), (0, 0), (1, 0), (2, 0), (2, 1)]


def _plan_image(w, H, W):
    host = {k: v.float().contiguous() for k, v in w.items() if v.is_floating_point()}
    arr = (TensorDesc * len(host))()
    for i, (k, t) in enumerate(host.items()):
        arr[i].name, arr[i].data, arr[i].numel = k.encode(), t.data_ptr(), t.numel()
    cfg = Config()
    cfg.classes, cfg.anchor_num, cfg.height, cfg.width, cfg.max_batch, cfg.device = 80, 3, H, W, 1, 0
    cap = 2 * OLD_IMG_FL + H16_FL + 64
    buf = np.zeros(cap, np.float32)
    name = C.create_string_buffer(256)
    n = _lib.lib().yfv2_debug_plan_image(C.byref(cfg), arr, len(host), 0, name, 256, buf.ctypes.data_as(C.c_void_p), cap)
    assert n >= 2 * OLD_IMG_FL + H16_FL and name.value.decode().startswith("stem"), (n, name.value)
    return buf[2 * OLD_IMG_FL:2 * OLD_IMG_FL + H16_FL]


def _decode(im):
    """-> W1, W2 as float32 [32 channels][4 groups][8 slots], shift*2^sw [32], 2^-sw"""
    u = im[:1024].view(np.uint32).reshape(2, 2, 64, 4)
    halves = np.stack((u & 0xffff, u >> 16), -1).astype(np.uint16).view(np.float16).astype(np.float32)   # [t][term][lane][d][e]
    wt = np.zeros((2, 32, 4, 8), np.float32)
    for t in range(2):
        for term in range(2):
            for l in range(64):
                wt[term, 16 * t + (l & 15), l >> 4] = halves[t, term, l].reshape(8)
    return wt[0], wt[1], im[1024:1056], float(im[1056])


def _split(x):
    h1 = x.astype(np.float16)
    h2 = (x - h1.astype(np.float32)).astype(np.float16)
    return h1.astype(np.float32), h2.astype(np.float32)


def _kernel_model(x, im, u8=False):
    """x (3, H, W) float32 -> (H/4, W/4, 24): the kernel's arithmetic, conv rows vectorised over the image.
    u8: stem_h3u_kernel - x holds the integers 0..255 (one exact fp16 term each), two products per MAC, its own constants."""
    w1, w2, shift, unscale = _decode(im)
    if u8:
        shift, unscale = im[1060:1092], float(im[1092])
    _, H, W = x.shape
    CH, CW = H // 2, W // 2
    xp = np.zeros((3, H + 2, W + 6), np.float32)       # row -1 and column -1 are padding; columns past W only feed zero weights
    xp[:, 1:H + 1, 1:W + 1] = x
    conv = np.zeros((32, CH, CW), np.float32)
    for parity in (0, 1):                               # tile E: conv columns 2px, tile O: 2px + 1
        n = (CW + 1 - parity) // 2
        pxs = np.arange(n)
        slots = np.zeros((4, 8, CH, n), np.float32)     # [g][slot][conv row][px]
        ys = np.arange(CH)
        for g in range(3):
            for j, (ky, kx) in enumerate(TAP):
                col = 4 * pxs + 2 * parity - 1 + kx     # input column of tap kx for conv column 2px + parity
                slots[g, j] = xp[g][(2 * ys - 1 + ky + 1)[:, None], (col + 1)[None, :]]
        for c, j in ((0, 2), (1, 3), (2, 7)):          # lane group 3: tap (2,2) of channel c (values fetched from lanes (p, c) by ds_bpermute)
            col = 4 * pxs + 2 * parity + 1
            slots[3, j] = xp[c][(2 * ys + 1 + 1)[:, None], (col + 1)[None, :]]
        # what group 3's other slots hold is data too (rows of other channels), but the filter is zero there: check that
        assert not w1[:, 3, [0, 1, 4, 5, 6]].any() and not w2[:, 3, [0, 1, 4, 5, 6]].any()
        x1, x2 = _split(slots if u8 else slots * np.float32(256.0))       # the kernel's exact 2^8 prescale of the image
        if u8:
            assert not x2.any() and (x1 == slots).all()  # a pixel is one fp16 term
        acc = np.broadcast_to(shift[:, None, None], (32, CH, n)).astype(np.float32).copy()
        for wa, xb in (((w2, x1), (w1, x1)) if u8 else ((w1, x2), (w2, x1), (w1, x1))):   # the kernel's product order; every product is exact in fp32
            acc = (acc.astype(np.float64) + np.einsum("cgj,gjyn->cyn", wa.astype(np.float64), xb.astype(np.float64))).astype(np.float32)
        conv[:, :, parity::2] = acc
    # max-pool 3x3 s2 p1 on the raw accumulators (0 stands for the padding: ReLU follows), then ReLU and the unscale
    cp = np.zeros((32, CH + 2, CW + 2), np.float32)
    cp[:, 1:-1, 1:-1] = conv
    pooled = np.zeros((32, CH // 2, CW // 2), np.float32)
    for dy in range(3):
        for dx in range(3):
            pooled = np.maximum(pooled, cp[:, dy:dy + CH:2, dx:dx + CW:2])
    return (np.maximum(pooled, 0) * np.float32(unscale))[:24].transpose(1, 2, 0)


def test_stem16_host_packing_and_dataflow_vs_oracle():
    for seed, (H, W) in ((3, (64, 96)), (4, (352, 352))):
        w = yfv2.random_state_dict(seed)
        w["backbone.first_conv.1.running_mean"] = torch.randn(24) * 0.1       # a BN that actually shifts and scales
        w["backbone.first_conv.1.running_var"] = torch.rand(24) + 0.5
        w["backbone.first_conv.1.weight"] = torch.randn(24)                   # negative scales too
        w["backbone.first_conv.1.bias"] = torch.randn(24) * 0.3
        im = _plan_image(w, H, W)
        torch.manual_seed(seed)
        x = torch.rand(1, 3, H, W)
        got = _kernel_model(x[0].numpy(), im)
        ref = oracle.forward_stages(w, x)["stem"][0].permute(1, 2, 0).numpy()
        assert got.shape == ref.shape
        err = np.abs(got - ref).max()
        assert err <= 2e-6 * max(1.0, np.abs(ref).max()), "stem16 dataflow model vs oracle: max abs err %g (max %g)" % (err, np.abs(ref).max())
        # uint8 pixels (stem_h3u_kernel): the oracle sees test.py:38's float() / 255
        xi = torch.randint(0, 256, (1, 3, H, W), generator=torch.Generator().manual_seed(seed), dtype=torch.uint8)
        got = _kernel_model(xi[0].numpy().astype(np.float32), im, u8=True)
        ref = oracle.forward_stages(w, xi.float() / 255.0)["stem"][0].permute(1, 2, 0).numpy()
        err = np.abs(got - ref).max()
        assert err <= 2e-6 * max(1.0, np.abs(ref).max()), "stem16 uint8 dataflow model vs oracle: max abs err %g (max %g)" % (err, np.abs(ref).max())


def test_fp16x3_is_as_accurate_as_the_fp32_convolution():
    """|error vs float64|: two-term fp16 operands, three products, fp32 accumulation - against a plain fp32 convolution of
    the same data (what the reference's ATen/oneDNN path and the fp32 MFMA compute).  Pixels in [0, 1], in [0, 255], and
    tiny ones (second terms in fp16's subnormal range)."""
    torch.manual_seed(0)
    w = torch.randn(24, 3, 3, 3) * 0.3
    for scale in (1.0, 255.0, 1.0 / 255.0, 1e-3):
        x = torch.rand(2, 3, 64, 64) * scale
        exact = F.conv2d(x.double(), w.double(), stride=2, padding=1)
        f32 = F.conv2d(x, w, stride=2, padding=1).double()
        sw = 14 - int(np.ceil(np.log2(float(w.abs().max()))))
        x1, x2 = _split(x.numpy() * np.float32(256.0)); w1, w2 = _split((w * 2.0 ** sw).numpy())   # the kernel's and the host's exact prescales
        up = 2.0 ** (sw + 8)
        terms = [F.conv2d(torch.from_numpy(a).double(), torch.from_numpy(b).double(), stride=2, padding=1) for a, b in ((x2, w1), (x1, w2), (x1, w1))]
        acc = torch.zeros_like(terms[0]).float()
        for t in terms:                                  # fp32 accumulation of exact products, term by term (coarser than the MFMA's per-product order: an upper bound)
            acc = (acc.double() + t).float()
        h3 = acc.double() / up
        e32, e16 = (f32 - exact).abs().max().item(), (h3 - exact).abs().max().item()
        assert e16 <= 2.0 * e32 + 1e-12 * scale, "scale %g: fp16x3 error %g vs fp32 conv error %g" % (scale, e16, e32)
