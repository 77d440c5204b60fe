#!/usr/bin/env python3
"""YFV2_LANES=N: one handle, ONE yfv2_detect call per batch of 256, the batch cut into N slices on N internal streams
(DESIGN.md section 5).  Per-step wall time of back-to-back calls on one caller stream, for N in the arguments (default 1 2 3 4),
next to the three-handle pipeline (bench.py's former `value` loop).  python tools/lanes_probe.py [N ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastestv2_amd as yfv2
dev = torch.device("cuda:0")
B = 256
NS = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
w = yfv2.random_state_dict(0)
anch = [12.64, 19.39, 37.88, 51.48, 55.71, 138.31, 126.91, 78.23, 131.57, 214.55, 279.92, 258.87]
g = torch.Generator(device=dev); g.manual_seed(1000)
x = torch.rand(B, 3, 352, 352, device=dev, generator=g)
engs = {}
for n in NS:
    os.environ["YFV2_LANES"] = str(n)
    e = yfv2.Engine(dev, 352, 352, 80, 3, max_batch=B); e.load_state_dict(w); e.set_anchors(anch); engs[n] = e
os.environ["YFV2_LANES"] = "1"
pipe = yfv2.DetectPipeline(dev, 352, 352, 80, 3, anchors=anch, max_batch=B, depth=3); pipe.load_state_dict(w)
def run(e, n, fwd_only=False):
    buf = e.new_det_buffers(B)
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(n):
        if fwd_only: e.forward(x)
        else: e.detect(x, 0.3, 0.4, out=buf)
    torch.cuda.synchronize(); return (time.time() - t0) / n
def run_pipe(n):
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(n): pipe.submit(x, 0.3, 0.4, wait_for_input=False)
    pipe.synchronize(); torch.cuda.synchronize(); return (time.time() - t0) / n
for e in engs.values(): run(e, 30)          # leave the idle clocks behind
ref = [t.clone() for t in engs[NS[0]].detect(x, 0.3, 0.4)]
for n, e in engs.items():
    got = e.detect(x, 0.3, 0.4)
    assert all(torch.equal(a, b) for a, b in zip(ref, got)), "lanes=%d differs" % n
for rep in range(3):
    for n, e in engs.items():
        dt = run(e, 48); df = run(e, 48, True)
        print("lanes %d: detect %.4f ms per call = %.1f k images/s; forward only %.4f ms = %.1f k images/s" % (n, 1e3 * dt, B / dt / 1e3, 1e3 * df, B / df / 1e3))
    run_pipe(6); dp = run_pipe(48)
    print("three handles / streams (DetectPipeline): detect %.4f ms per step = %.1f k images/s" % (1e3 * dp, B / dp / 1e3))
