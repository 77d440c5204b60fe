cd $GRAFT_REPO_ROOT
python tools/scale_probe.py 256 2>&1 | grep -v amdgpu | cut -c1-60,96-130
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o t -- python $GRAFT_REPO_ROOT/tools/scale_probe.py 256 > /dev/null 2>&1
f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1); python - <<PY
import csv
for r in list(csv.reader(open("$f")))[1:14]:
    if 'at::' in r[0]: continue
    print("%-60s %5s calls %8.1f us" % (r[0][:60], r[1], float(r[3])/1000))
PY
