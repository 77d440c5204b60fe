#!/usr/bin/env python3
"""What a plain streaming kernel reaches on this box (the yardstick for the strip kernels' GB/s): device-to-device copy,
read-only reduction and write-only fill of the stem's input size (381 MB) and of its traffic mix (read 381 MB + write 190 MB)."""
import os, sys, torch
dev = torch.device("cuda:0")
n = 256 * 3 * 352 * 352
x = torch.rand(n, device=dev); y = torch.empty_like(x); h = torch.empty(n // 2, device=dev)
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
B = n * 4
for name, f, nbytes in (("copy_ (hipMemcpyDtoD / copy kernel): read 381 MB + write 381 MB", lambda: y.copy_(x), 2 * B),
                        ("mul_ in place: read 381 MB + write 381 MB", lambda: x.mul_(1.0), 2 * B),
                        ("torch.mul(x, 1, out=y): read 381 MB + write 381 MB", lambda: torch.mul(x, 1.0, out=y), 2 * B),
                        ("sum: read 381 MB", lambda: x.sum(), B),
                        ("fill_: write 381 MB", lambda: y.fill_(1.0), B),
                        ("x[::2] -> h strided read 381 MB (half used) + write 190 MB", lambda: torch.add(x[0::2], x[1::2], out=h), B + B // 2)):
    s = t(f)
    print("%-70s %7.1f us  %6.2f TB/s" % (name, s * 1e6, nbytes / s / 1e12))
