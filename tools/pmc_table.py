#!/usr/bin/env python3
"""One table row per kernel from a rocprofv3 --pmc SQ pass (tools/gpu_pmc.sh pass 1): wait / issue-wait / active
shares of the wave cycles, MFMA busy share per SIMD, LDS bank-conflict share of the LDS-active cycles.
usage: tools/pmc_table.py gpurun_out/pmc1_TAG [profiles/TAG_pmc.json] > profiles/TAG_pmc_sq_summary.txt
With a second argument the same figures are also written as JSON together with tools/srchash.py's fingerprint of the tree:
bench.py quotes `mfma_busy` per kernel from the newest such file whose fingerprint matches the tree it runs from."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from srchash import source_hash

acc = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        per.setdefault((r["Dispatch_Id"], r["Kernel_Name"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for (_, k), cs in per.items():
        a = acc.setdefault(k, collections.defaultdict(list))
        for c, v in cs.items():
            a[c].append(v)
print("GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy fraction per SIMD = (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE) / 128\n")
print("%-58s %4s %11s %7s %7s %8s %9s %9s" % ("kernel", "n", "GUIact/n", "wait%", "instw%", "active%", "mfmaBusy%", "ldsConfl%"))
js = {"src_hash": source_hash(), "note": "per-kernel means of one rocprofv3 --pmc SQ pass; mfma_busy_pct = SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE / 128 (GRBM summed over 8 XCDs, busy cycles over 1024 SIMDs)", "kernels": {}}
for k, a in acc.items():
    m = {c: sum(v) / len(v) for c, v in a.items()}
    wc = max(m.get("SQ_WAVE_CYCLES", 1.0), 1.0)
    g = max(m.get("GRBM_GUI_ACTIVE", 1.0), 1.0)
    lds = max(m.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)
    print("%-58s %4d %11.0f %7.1f %7.1f %8.1f %9.1f %9.1f" % (k[:58], len(a["GRBM_GUI_ACTIVE"]), g, 100 * m.get("SQ_WAIT_ANY", 0) / wc,
          100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / g / 128,
          100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / lds))
    js["kernels"][k if len(k) < 100 else k[:97] + "..."] = {
        "n": len(a["GRBM_GUI_ACTIVE"]), "wait_pct": round(100 * m.get("SQ_WAIT_ANY", 0) / wc, 2), "inst_wait_pct": round(100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 2),
        "active_pct": round(100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 2), "mfma_busy_pct": round(100 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / g / 128, 2),
        "lds_conflict_pct": round(100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / lds, 2)}
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        json.dump(js, f, indent=1)
