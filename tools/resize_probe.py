#!/usr/bin/env python3
"""Throughput of yfv2_resize_u8 (cv2.resize INTER_LINEAR on the device) for a few frame sizes: frames/s and the
algorithmic HBM rate (source rows actually touched + output bytes).  usage: python tools/resize_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastestv2_amd as yfv2  # noqa: E402

dev = torch.device("cuda:0")
eng = yfv2.get_engine(dev, 352, 352)
for B, sh, sw in ((256, 480, 640), (256, 720, 1280), (64, 1080, 1920), (256, 240, 320)):
    frames = torch.randint(0, 256, (B, sh, sw, 3), dtype=torch.uint8, device=dev)
    out = torch.empty((B, 352, 352, 3), dtype=torch.uint8, device=dev)
    for _ in range(3):
        eng.resize(frames, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        eng.resize(frames, out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    rows_touched = min(sh, 2 * 352)                      # each output row blends two source rows; a reduction skips the rest
    bytes_alg = B * (rows_touched * sw * 3 + 352 * 352 * 3)
    print("resize %4dx%-4d -> 352x352  B=%3d  %.3f ms  %.0f frames/s  %.0f GB/s algorithmic" % (sw, sh, B, ms, B / ms * 1e3, bytes_alg / ms / 1e6))
