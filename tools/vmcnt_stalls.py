#!/usr/bin/env python3
"""Where does a kernel wait for ALL its outstanding memory requests right behind a request?  (The chain kernel of stage 3 lost 20 %
to this: loads or waits behind lane predicates leave paths on which the compiler assumes a load pending; it then guards the
destination registers where they are reused - s_waitcnt vmcnt(0) right behind the NEXT prefetch, a full load latency per phase.)
Disassembles the built objects' device code (make must have run) and lists, per kernel, every `s_waitcnt vmcnt(N)` with small N
that follows a global/buffer load by fewer than --near instructions.
usage: python tools/vmcnt_stalls.py [unit ...] [--near=40] [--max=1]"""
import os, re, subprocess, sys
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "yolo_fastestv2_amd", "csrc")
FLAGS = {"yfv2_stem": ["-fno-honor-nans", "-mllvm", "-amdgpu-mfma-vgpr-form=1"], "yfv2_stem16": ["-fno-honor-nans"],
         "yfv2_stage2": ["-fno-honor-nans", "-mllvm", "-amdgpu-mfma-vgpr-form=1"], "yfv2_stage2h": ["-fno-honor-nans"], "yfv2_towerh": ["-fno-honor-nans"],
         "yfv2_post": ["-ffp-contract=off"], "yfv2_pre": ["-ffp-contract=off"], "yfv2_loss": ["-ffp-contract=off"], "yfv2_train": ["-ffp-contract=off"]}
units = [a for a in sys.argv[1:] if not a.startswith("--")] or ["yfv2_block", "yfv2_towerh", "yfv2_stage2h", "yfv2_conv", "yfv2_stem16", "yfv2_post"]
near = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--near=")), 40))
nmax = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--max=")), 1))
for u in units:
    asm = "/tmp/vmcnt_%s.s" % u
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-fvisibility=hidden", "-Wno-unused-variable",
                    "-Wno-unused-but-set-variable", "-Wno-cuda-compat", *FLAGS.get(u, []), "-S", "--cuda-device-only", os.path.join(CSRC, u + ".hip"), "-o", asm],
                   capture_output=True)
    kern, n_since, depth, lines = None, 10 ** 9, 0, open(asm).read().splitlines()
    for ln in lines:
        t = ln.strip()
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kern = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]; n_since = 10 ** 9; depth = 0
            continue
        if not kern or not t or t.startswith(";") or t.startswith("."):
            if "Loop Header" in ln:
                d = re.search(r"Depth=(\d+)", ln); depth = int(d.group(1)) if d else depth
            continue
        if "Loop Header" in ln:
            d = re.search(r"Depth=(\d+)", ln); depth = int(d.group(1)) if d else depth
        if t.startswith("s_endpgm"):
            kern = None; continue
        if re.match(r"(global_load|buffer_load|flat_load)", t):
            n_since = 0; last = t.split()[0]; continue
        n_since += 1
        m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", t)
        if m and int(m.group(1)) <= nmax and n_since <= near:
            print("%-40s loop depth %d: %-28s %3d instructions after a %s" % (kern[:40], depth, t, n_since, last))
