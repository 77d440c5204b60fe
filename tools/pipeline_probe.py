#!/usr/bin/env python3
"""Does running consecutive bench steps on N handles / N streams raise the throughput (step i's decode + NMS launch and the
under-filled tail of every launch next to step i+1's launches)?  python tools/pipeline_probe.py [max handles]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastestv2_amd as yfv2
dev = torch.device("cuda:0")
B = 256
NMAX = int(sys.argv[1]) if len(sys.argv) > 1 else 4
w = yfv2.random_state_dict(0)
anch = [12.64, 19.39, 37.88, 51.48, 55.71, 138.31, 126.91, 78.23, 131.57, 214.55, 279.92, 258.87]
engs = []
for _ in range(NMAX):
    e = yfv2.Engine(dev, 352, 352, 80, 3, max_batch=B); e.load_state_dict(w); e.set_anchors(anch); engs.append(e)
g = torch.Generator(device=dev); g.manual_seed(1000)
x = torch.rand(B, 3, 352, 352, device=dev, generator=g)
streams = [torch.cuda.Stream(device=dev) for _ in range(NMAX)]
bufs = [e.new_det_buffers(B) for e in engs]
def run(n, nh, fwd_only=False):
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(n):
        k = i % nh
        with torch.cuda.stream(streams[k] if nh > 1 else torch.cuda.current_stream(dev)):
            if fwd_only: engs[k].forward(x)
            else: engs[k].detect(x, 0.3, 0.4, out=bufs[k])
    torch.cuda.synchronize(); return (time.time() - t0) / n
for rep in range(2):
    for nh in range(1, NMAX + 1):
        run(2 * nh, nh)
        dt = run(48, nh)
        run(2 * nh, nh, True)
        df = run(48, nh, True)
        print("%d handle(s) / stream(s): detect %.4f ms per step = %.1f k images/s; forward only %.4f ms = %.1f k images/s" % (nh, 1e3 * dt, B / dt / 1e3, 1e3 * df, B / df / 1e3))
