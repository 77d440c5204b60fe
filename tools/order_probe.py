#!/usr/bin/env python3
"""Does the order of measurement matter?  forward x N, detect x N, forward x N, detect x N ... on one engine (B=256, random weights):
python tools/order_probe.py [N]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastestv2_amd as yfv2
ANCHORS = [12.64, 19.39, 37.88, 51.48, 55.71, 138.31, 126.91, 78.23, 131.57, 214.55, 279.92, 258.87]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0"); B = 256
eng = yfv2.Engine(dev, 352, 352, 80, 3, anchors=ANCHORS, max_batch=B)
eng.load_state_dict(yfv2.random_state_dict(0))
x = torch.rand(B, 3, 352, 352, device=dev, generator=torch.Generator(device=dev).manual_seed(1000))
out = eng.new_det_buffers(B); lg = [torch.empty(s, device=dev) for s in eng.logit_shapes(B)]
def t(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for _ in range(3): eng.detect(x, 0.3, 0.4, out=out)
res = []
for rep in range(4):
    res.append(("detect", t(lambda: eng.detect(x, 0.3, 0.4, out=out), N)))
    res.append(("forward", t(lambda: eng.forward(x, out=lg), N)))
print("  ".join("%s %.1f" % r for r in res))
