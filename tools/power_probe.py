#!/usr/bin/env python3
"""Energy per launch.  bench.py's pipelined headline runs at ~1300 W of the package's 1400 W cap with the shader clock pulled down
to ~2.1 GHz: it is POWER-limited, so what a launch costs there is joules, not microseconds.  For every launch of the forward plan
(and the post launch through detect): repeat it alone for ~1 s (yfv2_debug_repeat_step), read the device's hwmon power sensor and
the in-kernel clock meanwhile, time it with events -> W, us, mJ per launch above the idle floor.
usage: python tools/power_probe.py [B]      (on the GPU box)"""
import glob
import os
import sys
import time
import ctypes as C

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastestv2_amd as yfv2  # noqa: E402
from yolo_fastestv2_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
props = torch.cuda.get_device_properties(dev)
pci = "%04x:%02x:%02x" % (int(getattr(props, "pci_domain_id", 0)), int(props.pci_bus_id), int(getattr(props, "pci_device_id", 0)))
hw = None
for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
    if pci in os.path.realpath(card).lower():
        cand = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
        if cand:
            hw = cand[0]
if hw is None:
    sys.exit("no hwmon node for %s" % pci)


def sensor():
    out = []
    for fn in ("power1_input", "power1_average"):
        try:
            with open(os.path.join(hw, fn)) as fh:
                out.append(float(fh.read()) * 1e-6)
                break
        except (OSError, ValueError):
            pass
    with open(os.path.join(hw, "freq1_input")) as fh:
        out.append(float(fh.read()) * 1e-6)
    return out


ANCHORS = [12.64, 19.39, 37.88, 51.48, 55.71, 138.31, 126.91, 78.23, 131.57, 214.55, 279.92, 258.87]
eng = yfv2.Engine(dev, 352, 352, 80, 3, anchors=ANCHORS, max_batch=B)
eng.load_state_dict(yfv2.random_state_dict(0))
x = torch.rand(B, 3, 352, 352, device=dev)
lg = [torch.empty(s, device=dev) for s in eng.logit_shapes(B)]
ptrs = (C.c_void_p * 6)(*[t.data_ptr() for t in lg])
stages = eng.stages()
ms = eng.profile_forward(x, iters=3)
L = _lib.lib()
stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
torch.cuda.synchronize()
time.sleep(1.0)
idle = sensor()
print("idle: %.0f W at %.0f MHz (hwmon %s)" % (idle[0], idle[1], hw))


def measure(enqueue, per_iter_ms, label, streams=None):
    streams = streams or [torch.cuda.current_stream(dev)]
    iters = max(50, int(1000.0 / max(per_iter_ms, 0.005)))
    enqueue(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enqueue(iters)                               # asynchronous: the device works through the queue while the host reads the sensor
    samples = []
    while not all(st.query() for st in streams):
        if time.perf_counter() - t0 > 0.25:      # the sensor averages over a window: skip the ramp
            samples.append(sensor())
        time.sleep(0.02)
    us = (time.perf_counter() - t0) / iters * 1e6
    torch.cuda.synchronize()
    if not samples:
        samples = [sensor()]
    w = sum(s[0] for s in samples) / len(samples)
    f = sum(s[1] for s in samples) / len(samples)
    print("%-64s %7.1f us  %6.0f W  %5.0f MHz  %7.2f mJ/launch (%6.2f above idle)  [%d samples]" % (label[:64], us, w, f, w * us * 1e-3, (w - idle[0]) * us * 1e-3, len(samples)))
    return us, w


tot_us = tot_mj = 0.0
for i, st in enumerate(stages):
    def enq(n, i=i):
        _lib.check(L.yfv2_debug_repeat_step(eng._h, C.c_void_p(x.data_ptr()), B, ptrs, i, n, stream), eng._h)
    us, w = measure(enq, ms[i], st["name"])
    tot_us += us; tot_mj += (w - idle[0]) * us * 1e-3
out = eng.new_det_buffers(B)


def enq_fwd(n):
    for _ in range(n):
        eng.forward(x, out=lg)


def enq_det(n):
    for _ in range(n):
        eng.detect(x, 0.3, 0.4, out=out, check=False)


print("sum of the launches: %.1f us, %.1f mJ above idle" % (tot_us, tot_mj))
fu, fw = measure(enq_fwd, sum(ms), "whole forward, one stream")
du, dw = measure(enq_det, sum(ms) + 0.07, "forward + decode + NMS, one stream")
print("post launch by difference: %.1f us, %.1f mJ above idle" % (du - fu, (dw - idle[0]) * du * 1e-3 - (fw - idle[0]) * fu * 1e-3))
pipe = yfv2.DetectPipeline(dev, 352, 352, 80, 3, anchors=ANCHORS, max_batch=B, depth=3)
pipe.load_state_dict(yfv2.random_state_dict(0))


def enq_pipe(n):
    for _ in range(n):
        with pipe.slot() as (_, e, bufs):
            e.detect(x, 0.3, 0.4, out=bufs, check=False)


measure(enq_pipe, 0.65, "forward + decode + NMS, three batches in flight", streams=pipe.streams)
