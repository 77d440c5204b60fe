#!/usr/bin/env python3
"""stem + stage2.0 as ONE launch (front_kernel, YFV2_FRONT=1) against the two launches: stage-2 activations and logits must be
BIT-identical (same instructions on the same operands in the same order); then the per-launch event times of both plans.
usage: python tools/front_probe.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yolo_fastestv2_amd as yfv2
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sd = yfv2.random_state_dict(0)


def engine(front, H, W, mb):
    os.environ["YFV2_FRONT"] = "1" if front else "0"
    try:
        e = yfv2.Engine(dev, H, W, 80, 3, max_batch=mb)
        e.load_state_dict(sd)          # (the plan is built when the weights arrive)
    finally:
        os.environ.pop("YFV2_FRONT", None)
    return e


CASES = ((352, 352, B),) if "quick" in sys.argv else ((352, 352, 5), (352, 352, B), (320, 320, 3), (288, 384, 2), (64, 96, 3), (512, 512, 2), (96, 1024, 2), (32, 32, 4), (352, 32, 2))
for (H, W, n) in CASES:
    x = torch.rand(n, 3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(H + n))
    e0, e1 = engine(False, H, W, n), engine(True, H, W, n)
    l0 = [t.clone() for t in e0.forward(x)]; a0 = e0.debug_activation(1, n)
    l1 = [t.clone() for t in e1.forward(x)]; a1 = e1.debug_activation(1, n)
    torch.cuda.synchronize()
    d = (a0 != a1)
    print("%dx%d B=%d: stage 2: %d of %d elements differ (max |d| %.3g); logits equal: %s; guard %d / %d" % (
        H, W, n, int(d.sum()), a0.numel(), float((a0 - a1).abs().max()), all(torch.equal(p, q) for p, q in zip(l0, l1)), e0.nonfinite(), e1.nonfinite()), flush=True)
    if int(d.sum()):
        idx = d.nonzero().flatten()[:8].tolist()
        per = a0.numel() // n
        print("   first differing (image, y, x, c):", [(i // per, (i % per) // 48 // (W // 8), (i % per) // 48 % (W // 8), i % 48) for i in idx])
    if (H, W, n) == (352, 352, B):
        for name, e in (("two launches", e0), ("front_kernel", e1)):
            for rep in range(2):
                ms = e.profile_forward(x, iters=5)
                st = e.stages()
                print("   %-13s #%d: %s | total %.1f us" % (name, rep, "  ".join("%.1f" % (1e3 * m) for m in ms[:6]), 1e3 * sum(ms)), flush=True)
    del e0, e1
