#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes (tools/gpu_traffic.sh): one calibration copy of a
known size (x.clone(): reads and writes 381 MB at B=256), two forwards, then the fused post launch in both regimes -
yfv2_detect on the bench's own batch (random-init weights: 300 kept boxes per image) and on COCO weights with the
JPEG-derived batch at 0.3 / 0.4 (a handful of detections per image)."""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
import yolo_fastestv2_amd as yfv2
B = 256
dev = torch.device("cuda:0")
eng = yfv2.Engine(dev, 352, 352, 80, 3, anchors=bench.ANCHORS, max_batch=B)
eng.load_state_dict(yfv2.random_state_dict(0))
g = torch.Generator(device=dev).manual_seed(1000)
x = torch.rand(B, 3, 352, 352, device=dev, generator=g)
torch.cuda.synchronize()
y = x.clone()            # calibration: 2 * B*3*352*352*4 bytes of pure streaming
torch.cuda.synchronize()
for _ in range(2):
    eng.forward(x)
torch.cuda.synchronize()
for _ in range(2):
    eng.detect(x, 0.3, 0.4)
torch.cuda.synchronize()
gold = os.path.join(REPO, "tests", "golden")
z = np.load(os.path.join(gold, "weights_coco.npz"))
eng_c = yfv2.Engine(dev, 352, 352, 80, 3, anchors=bench.ANCHORS, max_batch=B)
eng_c.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files})
xc = bench.batch_from_reference_images(np.load(os.path.join(gold, "images_u8.npz"))["images"], B, seed=3).to(dev)
for _ in range(2):
    eng_c.detect(xc, 0.3, 0.4)
torch.cuda.synchronize()
print("calibration_bytes_each_way", x.numel() * 4)
