#!/usr/bin/env python3
"""Maximum sizes: ONE call on a batch far beyond BASELINE.json's 256 images per GPU - tensors beyond 2^31 elements and beyond
4 GiB (the input, the stem's output, the stage-2 planes, the decoded tensor) - run in its OWN interpreter by
tests/test_gpu_parity.py.  The reference has no batch limit (its tensors are ATen's, utils/utils.py:251 loops over images); a
288 GB device holds such a batch, so every per-image base address in the kernels has to be 64-bit arithmetic.

Size-independent property (no oracle run on thousands of images): the batch is K copies of one 256-image block, and every copy's
logits, decoded rows and detections must be BIT-identical to what the 256-image block gives as a batch of its own (batch-position
invariance, tests/test_gpu_parity.py::test_batch_invariance_and_permutation, at a size where a 32-bit offset would wrap).
usage: large_batch.py B [uint8|fp32 [H W]]      (B a multiple of 256; environment switches such as YFV2_BF6=0 select the other plans)"""
import faulthandler
import os
import sys

faulthandler.enable()
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import yolo_fastestv2_amd as yfv2  # noqa: E402

ANCHORS = [12.64, 19.39, 37.88, 51.48, 55.71, 138.31, 126.91, 78.23, 131.57, 214.55, 279.92, 258.87]


def mark(msg):
    print("[large_batch] " + msg, flush=True)


def main():
    B = int(sys.argv[1])
    u8 = len(sys.argv) > 2 and sys.argv[2] == "uint8"
    H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (352, 352)
    assert B % 256 == 0 and B >= 512
    K = B // 256
    dev = torch.device("cuda:0")
    sd = yfv2.random_state_dict(11)
    g = torch.Generator(device=dev).manual_seed(21)
    blk = torch.rand(256, 3, H, W, device=dev, generator=g)
    if u8:
        blk = (blk.permute(0, 2, 3, 1) * 255.0).round().clamp(0, 255).to(torch.uint8).contiguous()

    small = yfv2.Engine(dev, H, W, 80, 3, anchors=ANCHORS, max_batch=256)
    small.load_state_dict(sd)
    ref_logits = [t.clone() for t in small.forward(blk)]
    ref_dec = small.decode(ref_logits).clone()
    ref_det = [t.clone() for t in small.detect(blk, 0.3, 0.4)]
    small.check_finite("the 256-image block")
    torch.cuda.synchronize()
    assert int(ref_det[2].min()) > 0, "the block yields no detections: the comparison below would be empty"
    mark("block of 256 done (%d detections)" % int(ref_det[2].sum()))

    big = yfv2.Engine(dev, H, W, 80, 3, anchors=ANCHORS, max_batch=B)
    big.load_state_dict(sd)
    x = blk.repeat(K, 1, 1, 1)
    mark("input %s: %.2f GiB, %d elements" % (tuple(x.shape), x.numel() * x.element_size() / 2 ** 30, x.numel()))
    logits = big.forward(x)
    torch.cuda.synchronize()
    mark("forward done")
    for name, t, r in zip(("reg2", "obj2", "cls2", "reg3", "obj3", "cls3"), logits, ref_logits):
        t = t.view(K, 256, *t.shape[1:])
        for k in range(K):
            assert torch.equal(t[k], r), "logits %s: copy %d (images %d..%d) differs from the 256-image batch" % (name, k, 256 * k, 256 * k + 255)
    mark("logits ok")
    dec = big.decode(logits)
    torch.cuda.synchronize()
    mark("decode done: %.2f GiB" % (dec.numel() * 4 / 2 ** 30))
    dv = dec.view(K, 256, *dec.shape[1:])
    for k in range(K):
        assert torch.equal(dv[k], ref_dec), "decoded rows: copy %d differs" % k
    del dec, dv
    mark("decode ok")
    det = big.detect(x, 0.3, 0.4)
    big.check_finite("the large batch")
    torch.cuda.synchronize()
    mark("detect done")
    for name, t, r in zip(("dets", "idx", "cnt"), det, ref_det):
        t = t.view(K, 256, *t.shape[1:])
        for k in range(K):
            if name == "cnt":
                assert torch.equal(t[k], r), "detection counts: copy %d differs" % k
            else:     # rows past an image's count are unspecified padding
                n = ref_det[2].view(256, *([1] * (r.dim() - 1)))
                live = torch.arange(r.shape[1], device=dev).view(1, -1, *([1] * (r.dim() - 2))) < n
                assert torch.equal(torch.where(live, t[k], torch.zeros_like(r)), torch.where(live, r, torch.zeros_like(r))), \
                    "%s: copy %d differs" % (name, k)
    mark("detections ok")
    print("LARGE BATCH OK B=%d%s %dx%d" % (B, " uint8" if u8 else "", H, W), flush=True)


if __name__ == "__main__":
    main()
