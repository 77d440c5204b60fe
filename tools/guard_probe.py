#!/usr/bin/env python3
"""What does the default plan do with activations beyond fp16's range, stage by stage?  (range guard, include/yfv2.h)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastestv2_amd as yfv2
from oracle import yfv2_oracle as oracle
dev = torch.device("cuda:0")
w = yfv2.random_state_dict(3)
x = torch.rand(3, 3, 352, 352, generator=torch.Generator().manual_seed(21))
big = {k: v.clone() for k, v in w.items()}
big["backbone.first_conv.1.weight"] *= 4000.0
big["backbone.first_conv.1.bias"] *= 4000.0
eng = yfv2.Engine(dev, 352, 352, 80, 3, max_batch=4)
eng.load_state_dict(big)
out = eng.forward(x.to(dev))
torch.cuda.synchronize()
ref = oracle.forward_stages(big, x)
for which, key in enumerate(("stem", "stage2", "c2", "c3", "s2", "s3")):
    r = ref[key].permute(0, 2, 3, 1).contiguous()
    got = eng.debug_activation(which, 3).reshape(r.shape)
    print("%-7s oracle max %12.1f   device max %12.1f  nan %d  inf %d  zeros %.3f (oracle zeros %.3f)  max abs diff %g" % (
        key, float(r.abs().max()), float(torch.nan_to_num(got, nan=0.0, posinf=0.0, neginf=0.0).abs().max()), int(torch.isnan(got).sum()), int(torch.isinf(got).sum()),
        float((got == 0).float().mean()), float((r == 0).float().mean()), float(torch.nan_to_num(got - r, nan=0.0).abs().max())))
print("logit nan counts", [int(torch.isnan(t).sum()) for t in out], "flag", eng.nonfinite())
