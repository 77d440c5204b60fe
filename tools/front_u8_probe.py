#!/usr/bin/env python3
"""uint8 input: front2_kernel<.., U8> against stem_h3u_kernel + s2h_kernel (YFV2_FRONT=0): stage 2 and logits bit-identical; times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yolo_fastestv2_amd as yfv2
dev = torch.device("cuda:0")
sd = yfv2.random_state_dict(0)
def engine(front, H, W, mb):
    os.environ["YFV2_FRONT"] = "1" if front else "0"
    try:
        e = yfv2.Engine(dev, H, W, 80, 3, max_batch=mb); e.load_state_dict(sd)
    finally:
        os.environ.pop("YFV2_FRONT", None)
    return e
for (H, W, n) in ((352, 352, 5), (352, 352, 256), (320, 320, 3), (288, 384, 2), (512, 512, 2), (96, 1024, 2), (416, 416, 3)):
    x = torch.randint(0, 256, (n, H, W, 3), device=dev, dtype=torch.uint8, generator=torch.Generator(device=dev).manual_seed(H + n))
    e0, e1 = engine(False, H, W, n), engine(True, H, W, n)
    l0 = [t.clone() for t in e0.forward(x)]; a0 = e0.debug_activation(1, n); s0 = e0.debug_activation(0, n)
    l1 = [t.clone() for t in e1.forward(x)]; a1 = e1.debug_activation(1, n); s1 = e1.debug_activation(0, n)
    torch.cuda.synchronize()
    print("%dx%d B=%d uint8: stage 2: %d of %d differ; stem (debug hook) equal: %s; logits equal: %s" % (H, W, n, int((a0 != a1).sum()), a0.numel(), torch.equal(s0, s1), all(torch.equal(p, q) for p, q in zip(l0, l1))), flush=True)
    if n == 256:
        for name, e in (("two launches", e0), ("front2 u8", e1)):
            for _ in range(5): e.forward(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50): e.forward(x)
            torch.cuda.synchronize(); print("   %-13s forward %.1f us" % (name, (time.perf_counter() - t0) / 50 * 1e6), flush=True)
