#!/usr/bin/env python3
"""Does a HIP graph help?  Engine.detect (14 forward launches + the post launch) captured into a graph (torch.cuda.CUDAGraph
captures whatever is launched on its stream - the library's kernels included) and replayed, against the same call issued eagerly,
at several batch sizes.  The forward is launch-bound only at small batches (0.35 ms for ONE image: 14 launches of per-image serial
work with a launch boundary between them).  usage: python tools/graph_probe.py [B ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastestv2_amd as yfv2  # noqa: E402

ANCHORS = [12.64, 19.39, 37.88, 51.48, 55.71, 138.31, 126.91, 78.23, 131.57, 214.55, 279.92, 258.87]
batches = [int(b) for b in sys.argv[1:]] or [1, 8, 32, 256]
dev = torch.device("cuda:0")
for B in batches:
    eng = yfv2.Engine(dev, 352, 352, 80, 3, anchors=ANCHORS, max_batch=B)
    eng.load_state_dict(yfv2.random_state_dict(0))
    x = torch.rand(B, 3, 352, 352, device=dev)
    out = eng.new_det_buffers(B)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        for _ in range(5):
            eng.detect(x, 0.3, 0.4, out=out, check=False)        # eager warm-up on the capture stream (function attributes, lazy set-up)
    torch.cuda.synchronize()
    ref = [t.clone() for t in out]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        eng.detect(x, 0.3, 0.4, out=out, check=False)
    for t in out:
        t.zero_()
    g.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(out, ref))
    n = 2000 if B <= 32 else 300

    def timeit(fn):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    def eager():
        with torch.cuda.stream(side):
            eng.detect(x, 0.3, 0.4, out=out, check=False)
    te = timeit(eager)
    tg = timeit(g.replay)
    print("B=%4d: eager %8.1f us per call, graph replay %8.1f us (%+.1f %%), results %s" % (B, te, tg, 100.0 * (tg - te) / te, "bit-identical" if same else "DIFFER"))
    del g, eng
