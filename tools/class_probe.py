#!/usr/bin/env python3
"""Step-by-step probe of a non-default class count (prints a marker after every synchronised stage)."""
import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yolo_fastestv2_amd as yfv2
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
print("import ok", flush=True)
eng = yfv2.Engine(dev, 352, 352, nc, 3, anchors=[12.64, 19.39, 37.88, 51.48, 55.71, 138.31, 126.91, 78.23, 131.57, 214.55, 279.92, 258.87], max_batch=2)
print("create ok", flush=True)
eng.load_state_dict(yfv2.random_state_dict(7, classes=nc))
torch.cuda.synchronize(); print("weights ok", flush=True)
x = torch.rand(2, 3, 352, 352, device=dev)
out = eng.forward(x); torch.cuda.synchronize(); print("forward ok", [tuple(o.shape) for o in out], flush=True)
dec = eng.decode(out); torch.cuda.synchronize(); print("decode ok", tuple(dec.shape), flush=True)
d, i, c = eng.nms(dec, 0.3, 0.4); torch.cuda.synchronize(); print("nms ok", c.tolist(), flush=True)
d, i, c = eng.detect(x, 0.3, 0.4); torch.cuda.synchronize(); print("detect ok", c.tolist(), flush=True)
