#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes (tools/gpu_traffic.sh): one calibration copy of a
known size (x.clone(): reads and writes 381 MB at B=256) followed by two forwards."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastestv2_amd as yfv2
B = 256
dev = torch.device("cuda:0")
eng = yfv2.Engine(dev, 352, 352, 80, 3, max_batch=B)
eng.load_state_dict(yfv2.random_state_dict(0))
x = torch.rand(B, 3, 352, 352, device=dev)
torch.cuda.synchronize()
y = x.clone()            # calibration: 2 * B*3*352*352*4 bytes of pure streaming
torch.cuda.synchronize()
for _ in range(2):
    eng.forward(x)
torch.cuda.synchronize()
print("calibration_bytes_each_way", x.numel() * 4)
