#!/bin/bash
# LDS counters of the tower launches (one PMC pass over a forward-only run): conflict share, LDS-active cycles, instruction counts
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6_ldspmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in cur prev; do
  [ $v = prev ] && export YFV2_LIB=$ROOT/yolo_fastestv2_amd/libyfv2_prev.so
  timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/v$v -o x -- python $ROOT/tools/fwd_only.py > $OUT/v$v.log 2>&1; echo "rc=$?"
  python - <<PY
import csv, glob, collections
acc = collections.OrderedDict()
for f in glob.glob("$OUT/v$v/**/*counter_collection.csv", recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        per.setdefault((r["Dispatch_Id"], r["Kernel_Name"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for (_, k), cs in per.items():
        a = acc.setdefault(k, collections.defaultdict(list))
        for c, val in cs.items(): a[c].append(val)
for k, a in acc.items():
    if not any(t in k for t in ("tower", "chain", "pool", "s3h", "s4h", "front2")): continue
    m = {c: sum(x) / len(x) for c, x in a.items()}
    print("%-40s n=%d" % (k[:40], len(a["GRBM_GUI_ACTIVE"])), {c: round(x) for c, x in m.items()}, "conflict %.1f %%" % (100 * m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1)))
PY
done
