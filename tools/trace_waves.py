#!/usr/bin/env python3
"""Per-wave cycle stamps (workgroup 0) of ONE launch of the forward plan:  python tools/trace_waves.py STEP [B]
STEP = index in the plan (see tools/scale_probe.py for the order) or a substring of the step name.  Prints, per wave, the stamps in cycles since the
earliest stamp 0 of the workgroup; kernels with stamps: block_s1w / block_s1x2 / tower2 (YFV2_WSTAMP in yfv2_block.hip)."""
import ctypes as C, os, sys
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastestv2_amd as yfv2
from yolo_fastestv2_amd import _lib
dev = torch.device("cuda:0")
if sys.argv[1].isdigit():
    step = int(sys.argv[1])
else:   # a substring of the step's name
    probe = yfv2.Engine(dev, 352, 352, 80, 3, max_batch=1)
    probe.load_state_dict(yfv2.random_state_dict(0))
    step = [i for i, st in enumerate(probe.stages()) if sys.argv[1] in st["name"]][0]
    del probe
os.environ["YFV2_TRACE"] = "1"; os.environ["YFV2_TRACE_STEP"] = str(step)
eng = yfv2.Engine(dev, 352, 352, 80, 3, max_batch=B)
eng.load_state_dict(yfv2.random_state_dict(0))
print("step %d: %s" % (step, eng.stages()[step]["name"]))
x = torch.rand(B, 3, 352, 352, device=dev)
for _ in range(3):
    eng.forward(x)
torch.cuda.synchronize()
N = 64 + 32 * 16
buf = torch.zeros(2 * N, dtype=torch.float32)
_lib.lib().yfv2_debug_activation(eng._h, 100, B, C.c_void_p(buf.data_ptr()), 2 * N)
st = buf.view(torch.int64).tolist()
waves = [st[64 + 32 * w: 64 + 32 * w + 32] for w in range(16)]
waves = [w for w in waves if w[0] != 0]
t0 = min(w[0] for w in waves)
nst = max(i for w in waves for i, v in enumerate(w) if v != 0) + 1
print("wave " + "".join("%8d" % i for i in range(nst)))
for k, w in enumerate(waves):
    print("%4d " % k + "".join("%8d" % (v - t0 if v else -1) for v in w[:nst]))
