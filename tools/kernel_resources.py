#!/usr/bin/env python3
"""Register / spill table of every kernel of one or more translation units (hipcc -Rpass-analysis=kernel-resource-usage):
usage: python tools/kernel_resources.py yfv2_block yfv2_stage2h ... [--filter substring]"""
import re, subprocess, sys, os
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "yolo_fastestv2_amd", "csrc")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
flt = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--filter=")]
for unit in args:
    extra = ["-ffp-contract=off"] if unit in ("yfv2_post", "yfv2_loss", "yfv2_train") else []
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-fvisibility=hidden", "-Wno-unused-variable",
           "-Wno-unused-but-set-variable", "-Wno-cuda-compat", *extra, "-c", os.path.join(CSRC, unit + ".hip"), "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = t.split(":", 1)[1].strip(); rows[cur] = {}
        elif cur and ":" in t:
            k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
    for name, r in rows.items():
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem).replace("void ", "")
        if flt and not any(f in dem for f in flt):
            continue
        print("%-46s VGPR %3s AGPR %3s SGPR %3s  spill V %s S %s  scratch %s  occupancy %s  LDS %s" % (
            dem[:46], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"),
            r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
