#!/usr/bin/env python3
"""Does running consecutive bench steps on TWO handles / two streams raise the throughput (step i's decode + NMS launch next
to step i+1's first launches)?  python tools/pipeline_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastestv2_amd as yfv2
dev = torch.device("cuda:0")
B = 256
w = yfv2.random_state_dict(0)
anch = [12.64, 19.39, 37.88, 51.48, 55.71, 138.31, 126.91, 78.23, 131.57, 214.55, 279.92, 258.87]
engs = []
for _ in range(2):
    e = yfv2.Engine(dev, 352, 352, 80, 3, max_batch=B); e.load_state_dict(w); e.set_anchors(anch); engs.append(e)
g = torch.Generator(device=dev); g.manual_seed(1000)
x = torch.rand(B, 3, 352, 352, device=dev, generator=g)
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
bufs = [e.new_det_buffers(B) for e in engs]
def run(n, two):
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(n):
        k = i & 1 if two else 0
        with torch.cuda.stream(streams[k] if two else torch.cuda.current_stream(dev)):
            engs[k].detect(x, 0.3, 0.4, out=bufs[k])
    torch.cuda.synchronize(); return (time.time() - t0) / n
for two in (False, True, False, True):
    run(4, two)
    dt = run(40, two)
    print("%s: %.4f ms per step, %.1f k images/s" % ("two handles, two streams" if two else "one handle, one stream ", 1e3 * dt, B / dt / 1e3))
