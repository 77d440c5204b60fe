#!/usr/bin/env python3
"""What a byte of plain HBM streaming costs in joules on this box: torch's own copy / reduce / fill kernels over the stem's
byte counts (381 MB in, 190 MB out), each repeated for ~1 s while the device's hwmon power sensor is read (tools/power_probe.py's
method).  The answer prices the HBM share of every launch's energy (is the power-limited pipelined headline paying
for bytes or for instructions?).   usage: python tools/traffic_energy_probe.py   (on the GPU box)"""
import glob
import os
import sys
import time

import torch

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
    for fn in ("power1_input", "power1_average"):
        try:
            with open(os.path.join(hw, fn)) as fh:
                p = float(fh.read()) * 1e-6
                break
        except (OSError, ValueError):
            pass
    with open(os.path.join(hw, "freq1_input")) as fh:
        return p, float(fh.read()) * 1e-6


x = torch.rand(256, 3, 352, 352, device=dev)          # 380.6 MB
y = torch.empty_like(x)
h = torch.empty(128, 3, 352, 352, device=dev)         # 190.3 MB
acc = torch.zeros((), device=dev)
torch.cuda.synchronize(); time.sleep(1.0)
idle = sensor()
print("idle: %.0f W at %.0f MHz" % idle)


def measure(fn, nbytes_r, nbytes_w, label, est_us):
    iters = max(200, int(1.2e6 / est_us))
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    samples = []
    st = torch.cuda.current_stream(dev)
    while not st.query():
        if time.perf_counter() - t0 > 0.3:
            samples.append(sensor())
        time.sleep(0.02)
    us = (time.perf_counter() - t0) / iters * 1e6
    torch.cuda.synchronize()
    samples = samples or [sensor()]
    w = sum(s[0] for s in samples) / len(samples); f = sum(s[1] for s in samples) / len(samples)
    mj = (w - idle[0]) * us * 1e-3
    nb = nbytes_r + nbytes_w
    print("%-58s %7.1f us %6.0f W %5.0f MHz  %6.1f mJ above idle  %5.2f TB/s  %5.1f pJ/byte = %4.2f pJ/bit  [%d samples]" % (
        label, us, w, f, mj, nb / us * 1e-6, mj * 1e9 / nb, mj * 1e9 / nb / 8, len(samples)))


n = x.numel() * 4
measure(lambda: y.copy_(x), n, n, "copy 381 MB -> 381 MB (read + write)", 170)
measure(lambda: torch.sum(x.view(-1), dim=(0,), out=acc), n, 0, "sum of 381 MB (read only, one add per element)", 80)
measure(lambda: y.fill_(1.0), 0, n, "fill 381 MB (write only)", 70)
measure(lambda: h.copy_(x[:128]), n // 2, n // 2, "copy 190 MB -> 190 MB", 90)
measure(lambda: torch.mul(x, 1.5, out=y), n, n, "y = 1.5 x (read + write, one multiply per element)", 170)
