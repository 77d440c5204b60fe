#!/usr/bin/env python3
"""One process = one YFV2_VARIANT (the switch is read once per process): a fingerprint of the six logit maps on fixed inputs at a few
sizes, then the per-launch event times at B = 256.  Two runs with different variants must print the same fingerprints if the
variants are meant to be bit-identical.   usage: YFV2_VARIANT=n python tools/variant_check.py [pattern]"""
import hashlib, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yolo_fastestv2_amd as yfv2
dev = torch.device("cuda:0")
pat = sys.argv[1] if len(sys.argv) > 1 else "."
sd = yfv2.random_state_dict(0)
for (H, W, n) in ((352, 352, 7), (320, 320, 3), (288, 384, 2), (416, 416, 2), (512, 512, 2), (96, 1024, 2), (352, 352, 300)):
    x = torch.rand(n, 3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(H + n))
    e = yfv2.Engine(dev, H, W, 80, 3, max_batch=n); e.load_state_dict(sd)
    out = e.forward(x); torch.cuda.synchronize()
    h = hashlib.sha256()
    for t in out:
        h.update(t.cpu().numpy().tobytes())
    print("fingerprint %dx%d B=%d: %s  guard %d" % (H, W, n, h.hexdigest()[:16], e.nonfinite()), flush=True)
    del e
x = torch.rand(256, 3, 352, 352, device=dev)
e = yfv2.Engine(dev, 352, 352, 80, 3, max_batch=256); e.load_state_dict(sd)
st = e.stages()
for rep in range(2):
    ms = e.profile_forward(x, iters=5)
    print("   ".join("%s %.1f" % (s["name"].split(":")[0].replace("backbone.", "")[:14], 1e3 * m) for s, m in zip(st, ms) if re.search(pat, s["name"])) + "  | total %.1f us" % (1e3 * sum(ms)), flush=True)
