#!/usr/bin/env python3
"""Summarise the two HBM-traffic PMC passes of tools/gpu_traffic.sh into JSON.
Per MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
exactly half the bytes of wide (16 B/lane) coalesced reads -> doubled here.  The run contains a
calibration copy of known size (x.clone()) which must come out at ~1.0/1.0.  `src_hash` = tools/srchash.py of the tree
the passes ran on (bench.py quotes these figures only for the same tree).
usage: tools/traffic_summary.py gpurun_out/fetch_TAG gpurun_out/write_TAG KNOWN_BYTES > profiles/TAG_traffic.json
       (a directory is searched for *counter_collection.csv; a csv path works too)"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from srchash import source_hash


def csvs(path):
    return [path] if path.endswith(".csv") else glob.glob(path + "/**/*counter_collection.csv", recursive=True)


def per_kernel(path, ctr):
    d = collections.defaultdict(list)
    for f in csvs(path):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                d[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {k: (sum(x for _, x in v) / len(v), len(v), [x for _, x in sorted(v)]) for k, v in d.items()}


fe, wr = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
known = float(sys.argv[3])
out = {"units": "bytes per launch (mean over the launches of the pass); read = 2*FETCH_SIZE*1024 (gfx950 wide-read correction), write = WRITE_SIZE*1024",
       "src_hash": source_hash(), "workload": "tools/traffic_probe.py: B = 256, seeded random-init weights, torch.rand input; "
       "forward x2, then yfv2_detect on the same batch (300 kept boxes per image) and on COCO weights + the JPEG-derived batch (0.3 / 0.4)",
       "kernels": {}}
for k in fe:
    name = k if len(k) < 100 else k[:97] + "..."
    rd, wt = 2 * fe[k][0] * 1024, wr.get(k, (0.0, 0, []))[0] * 1024
    out["kernels"][name] = {"read_bytes": rd, "write_bytes": wt, "total_bytes": rd + wt, "launches": fe[k][1]}
    if "nms_kernel" in k and k in wr and len(fe[k][2]) == len(wr[k][2]):   # the post launch runs in two regimes: per launch, in dispatch order
        out["kernels"][name]["per_launch_total_bytes_in_dispatch_order"] = [2 * a * 1024 + b * 1024 for a, b in zip(fe[k][2], wr[k][2])]
cal = out["kernels"].get("__amd_rocclr_copyBuffer")
if cal:
    out["calibration"] = {"kernel": "__amd_rocclr_copyBuffer (x.clone())", "known_bytes_each_way": known,
                          "read_ratio": cal["read_bytes"] / known, "write_ratio": cal["write_bytes"] / known}
print(json.dumps(out, indent=1))
