"""Per-kernel means of rocprofv3 --pmc counter CSVs (one or more output dirs)."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per_dispatch = collections.defaultdict(dict)
        for row in csv.DictReader(open(f)):
            per_dispatch[(row["Dispatch_Id"], row["Kernel_Name"])][row["Counter_Name"]] = float(row["Counter_Value"])
        for (did, k), cs in per_dispatch.items():
            for c, v in cs.items():
                acc[k][c].append(v)
names = sorted({c for k in acc for c in acc[k]})
for k in acc:
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    print("\n" + k[:90] + "  (n=%d)" % len(next(iter(acc[k].values()))))
    for c in names:
        if c in m:
            print("   %-28s %16.0f" % (c, m[c]))
    g = m.get("GRBM_GUI_ACTIVE")
    if g and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        print("   -> MFMA busy per SIMD = %.1f %%   wait%% = %.1f   inst-wait%% = %.1f" % (100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / g / 128,
              100 * m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), 100 * m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)))
