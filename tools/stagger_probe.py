#!/usr/bin/env python3
"""Three handles / streams, 20 detect steps from an idle device (bench.py's timed region): does it matter how the first
three steps are spaced?  Enqueued back to back they start the same launches at the same time and stay in lockstep for tens of
steps; a host-side pause of a third of a step before the second and the third puts them out of phase from the start."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastestv2_amd as yfv2
dev = torch.device("cuda:0")
B, K = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 20
anch = [12.64, 19.39, 37.88, 51.48, 55.71, 138.31, 126.91, 78.23, 131.57, 214.55, 279.92, 258.87]
pipe = yfv2.DetectPipeline(dev, 352, 352, 80, 3, anchors=anch, max_batch=B, depth=3)
pipe.load_state_dict(yfv2.random_state_dict(0))
g = torch.Generator(device=dev); g.manual_seed(1000)
x = torch.rand(B, 3, 352, 352, device=dev, generator=g)
def run(n, pause_us):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        if pause_us and 1 <= i <= 2:
            t1 = time.perf_counter()
            while (time.perf_counter() - t1) * 1e6 < pause_us: pass
        pipe.submit(x, 0.3, 0.4, wait_for_input=False)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for _ in range(6): pipe.submit(x, 0.3, 0.4, wait_for_input=False)
for rep in range(3):
    for pause in (0, 150, 250, 350, 500):
        print("K = %d, pause %3d us before steps 1 and 2: %.4f ms per step" % (K, pause, 1e3 * run(K, pause)))
