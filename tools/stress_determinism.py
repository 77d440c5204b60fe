#!/usr/bin/env python3
"""Race hunt: the same input through the forward (and the fused post launch) many times, at several batch sizes, interleaved;
every repetition must reproduce the first one bit for bit (a data race or a missing barrier shows up as a flipped bit
sooner or later).  python tools/stress_determinism.py [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastestv2_amd as yfv2
ANCHORS = [12.64, 19.39, 37.88, 51.48, 55.71, 138.31, 126.91, 78.23, 131.57, 214.55, 279.92, 258.87]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
bad = 0
for (H, W) in ((352, 352), (320, 320), (288, 384)):
    eng = yfv2.Engine(dev, H, W, 80, 3, anchors=ANCHORS, max_batch=256)
    eng.load_state_dict(yfv2.random_state_dict(3))
    g = torch.Generator(device=dev).manual_seed(7)
    xs = {B: torch.rand(B, 3, H, W, device=dev, generator=g) for B in (256, 37, 1)}
    ref = {}
    for r in range(reps):
        for B, x in xs.items():
            out = [t.clone() for t in eng.forward(x)]
            d, i, c = [t.clone() for t in eng.detect(x, 0.3, 0.4)]
            cur = out + [d, i, c]
            if B not in ref:
                ref[B] = cur
            else:
                for k, (a, b) in enumerate(zip(ref[B], cur)):
                    if not torch.equal(a, b):
                        bad += 1
                        print("MISMATCH %dx%d B=%d rep %d tensor %d: %d elements differ" % (H, W, B, r, k, int((a != b).sum())))
    torch.cuda.synchronize()
    print("%dx%d: %d repetitions x 3 batch sizes done" % (H, W, reps))
print("stress: %s" % ("FAILED (%d mismatches)" % bad if bad else "bit-identical throughout"))
sys.exit(1 if bad else 0)
