#!/usr/bin/env python3
"""Per-launch duration of the forward plan at several batch sizes (hipEvent pairs,
yfv2_profile_forward).  A launch whose time barely moves with B is dominated by fixed
cost (prologue, latency chains); one that scales with B is throughput-bound.
usage: python tools/scale_probe.py [B ...]   (on the GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastestv2_amd as yfv2  # noqa: E402

batches = [int(b) for b in sys.argv[1:]] or [32, 64, 128, 256]
dev = torch.device("cuda:0")
eng = yfv2.Engine(dev, 352, 352, 80, 3, max_batch=max(batches))
eng.load_state_dict(yfv2.random_state_dict(0))
stages = eng.stages()
res = {}
for B in batches:
    x = torch.rand(B, 3, 352, 352, device=dev)
    eng.profile_forward(x, iters=2)
    res[B] = eng.profile_forward(x, iters=5)
print("%-96s" % "launch" + "".join("%9s" % ("B=%d" % b) for b in batches) + "   us/img@max")
for i, st in enumerate(stages):
    row = [res[b][i] * 1e3 for b in batches]
    print("%-96s" % st["name"][:95] + "".join("%9.1f" % v for v in row) + "   %8.3f" % (row[-1] / batches[-1]))
print("%-96s" % "TOTAL (us)" + "".join("%9.1f" % (sum(res[b]) * 1e3) for b in batches))
