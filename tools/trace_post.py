#!/usr/bin/env python3
"""Cycle stamps of the fused decode + NMS launch (workgroup 0 = image 0, thread 0):  python tools/trace_post.py [random|coco] [conf]
phases: decode | filter | sort | geometry | greedy | output"""
import ctypes as C, os, sys
os.environ["YFV2_TRACE"] = "1"; os.environ["YFV2_TRACE_STEP"] = "-2"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastestv2_amd as yfv2
from yolo_fastestv2_amd import _lib
ANCHORS = [12.64, 19.39, 37.88, 51.48, 55.71, 138.31, 126.91, 78.23, 131.57, 214.55, 279.92, 258.87]
which = sys.argv[1] if len(sys.argv) > 1 else "random"; conf = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
dev = torch.device("cuda:0"); B = 256
eng = yfv2.Engine(dev, 352, 352, 80, 3, anchors=ANCHORS, max_batch=B)
if which == "coco":
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "weights_coco.npz"))
    eng.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files})
else:
    eng.load_state_dict(yfv2.random_state_dict(0))
x = torch.rand(B, 3, 352, 352, device=dev, generator=torch.Generator(device=dev).manual_seed(1000))
out = eng.new_det_buffers(B)
for _ in range(3):
    eng.detect(x, conf, 0.4, out=out)
torch.cuda.synchronize()
buf = torch.zeros(256, dtype=torch.float32)
_lib.lib().yfv2_debug_activation(eng._h, 100, B, C.c_void_p(buf.data_ptr()), 256)
st = buf.view(torch.int64).tolist()
names = ["decode", "filter", "sort", "geometry", "greedy", "output"]
print("greedy: %d chunks of 64; thread 0: tests %d, scalar walk %d, barriers (= waiting for the slowest wave) %d ticks" % (st[9], st[10], st[11], st[12]))
print("%s weights, conf %.2f: n = %d candidates, kept %d; ticks per phase: %s; total %d" % (
    which, conf, st[7], st[8], "  ".join("%s %s" % (nm, st[i + 1] - st[i] if st[i + 1] >= st[i] > 0 else "-") for i, nm in enumerate(names)),   # (an image without candidates leaves early: its later stamps are not written)
    max(v for v in st[:7] if v > 0) - st[0]))
