#!/usr/bin/env python3
"""Where a batch far beyond 256 images first departs from the 256-image result: per stage activation (first 256 images) and per
logit map / 256-image copy.  usage: large_batch_diag.py B [B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yolo_fastestv2_amd as yfv2
ANCHORS = [12.64, 19.39, 37.88, 51.48, 55.71, 138.31, 126.91, 78.23, 131.57, 214.55, 279.92, 258.87]
dev = torch.device("cuda:0")
sd = yfv2.random_state_dict(11)
g = torch.Generator(device=dev).manual_seed(21)
blk = torch.rand(256, 3, 352, 352, device=dev, generator=g)
small = yfv2.Engine(dev, 352, 352, 80, 3, anchors=ANCHORS, max_batch=256); small.load_state_dict(sd)
ref = [t.clone() for t in small.forward(blk)]
ref_act = [small.debug_activation(w, 256) for w in range(6)]
names = ("reg2", "obj2", "cls2", "reg3", "obj3", "cls3")
for B in [int(v) for v in sys.argv[1:]] or [6144]:
    K = B // 256
    big = yfv2.Engine(dev, 352, 352, 80, 3, anchors=ANCHORS, max_batch=B); big.load_state_dict(sd)
    x = blk.repeat(K, 1, 1, 1)
    out = big.forward(x); torch.cuda.synchronize()
    print("B=%d" % B, flush=True)
    for w, nm in enumerate(("stem", "stage2", "C2", "C3", "S2", "S3")):
        a = big.debug_activation(w, 256)
        d = (a != ref_act[w])
        per = a.numel() // 256
        bad_imgs = d.view(256, per).any(1).nonzero().flatten().tolist()
        print("  act %-6s first 256 images: %d of %d elements differ; images %s" % (nm, int(d.sum()), a.numel(), bad_imgs[:12]), flush=True)
    for nm, t, r in zip(names, out, ref):
        t = t.view(K, 256, -1); r = r.view(256, -1)
        bad = []
        for k in range(K):
            d = (t[k] != r)
            if bool(d.any()):
                imgs = d.any(1).nonzero().flatten().tolist()
                bad.append((k, int(d.sum()), imgs[:6], len(imgs)))
        print("  logits %s: %d of %d copies differ %s" % (nm, len(bad), K, bad[:6]), flush=True)
    del big, x, out
    torch.cuda.empty_cache()
