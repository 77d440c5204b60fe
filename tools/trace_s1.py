#!/usr/bin/env python3
"""YFV2_TRACE=1 python tools/trace_s1.py : cycle stamps (workgroup 0, thread 0) of the LAST launch of the
block_s1 instantiation selected by YFV2_TRACE_C2 (default C2=48: stage3.7)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["YFV2_TRACE"] = "1"
import yolo_fastestv2_amd as yfv2
from yolo_fastestv2_amd import _lib
dev = torch.device("cuda:0")
for B in (16, 256):
    eng = yfv2.Engine(dev, 352, 352, 80, 3, max_batch=B)
    eng.load_state_dict(yfv2.random_state_dict(0))
    x = torch.rand(B, 3, 352, 352, device=dev)
    for _ in range(3):
        eng.forward(x)
    torch.cuda.synchronize()
    buf = torch.zeros(128, dtype=torch.float32)
    _lib.lib().yfv2_debug_activation(eng._h, 100, B, C.c_void_p(buf.data_ptr()), 128)
    st = buf.view(torch.int64).tolist()
    d = [st[i] - st[0] for i in range(7)]
    print("   staged+committed=%d  A-reads-done(last pass)=%d" % (st[8] - st[0], st[9] - st[0]))
    print("B=%d stamps (cycles since entry): copy_issued=%d prologue_done=%d waveA=%d phaseA=%d waveB=%d end=%d" % (B, d[1], d[2], d[3], d[4], d[5], d[6]))
