#!/bin/bash
# kernel start/end timestamps of one forward (does anything overlap across the handle's streams?)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/ovl -o ovl -- python $ROOT/tools/order_probe.py 3 > $OUT/ovl.log 2>&1; echo rc=$?
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/ovl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last forward: find the last stem kernel
idx = [i for i, r in enumerate(rows) if "stem_px" in r["Kernel_Name"]][-1]
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:idx + 22]:
    print("%-60s q%-3s start %8.1f  end %8.1f us" % (r["Kernel_Name"][:60], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3))
PY
