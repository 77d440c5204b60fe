#!/usr/bin/env python3
"""Time of the fused detect's decode + NMS legs at B=256 with the bench's inputs (random weights: 300 detections per image,
the NMS worst case) and with few candidates (conf 0.95):  python tools/nms_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastestv2_amd as yfv2
ANCHORS = [12.64, 19.39, 37.88, 51.48, 55.71, 138.31, 126.91, 78.23, 131.57, 214.55, 279.92, 258.87]
dev = torch.device("cuda:0"); B = 256
eng = yfv2.Engine(dev, 352, 352, 80, 3, anchors=ANCHORS, max_batch=B)
eng.load_state_dict(yfv2.random_state_dict(0))
x = torch.rand(B, 3, 352, 352, device=dev, generator=torch.Generator(device=dev).manual_seed(1000))
out = eng.new_det_buffers(B); lg = [torch.empty(s, device=dev) for s in eng.logit_shapes(B)]
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
f = t(lambda: eng.forward(x, out=lg))
for ct in (0.3, 0.95):
    d = t(lambda: eng.detect(x, ct, 0.4, out=out))
    print("conf %.2f: forward %.1f us, detect %.1f us -> decode + NMS %.1f us, mean detections %.1f" % (ct, f, d, d - f, float(out[2].float().mean())))
dec = eng.decode(eng.forward(x))
n = t(lambda: eng.nms(dec, 0.3, 0.4, out=out))
print("three-call NMS on the (B,1815,85) tensor: %.1f us" % n)
