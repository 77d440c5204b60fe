#!/bin/bash
# HBM traffic of every launch: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC has 4
# slots; MI355X_MICROARCH.md "HBM"), kernel-trace only.  usage: bash tools/gpu_traffic.sh TAG
TAG=${1:-traffic}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_$TAG -o $TAG -- python $ROOT/tools/traffic_probe.py > $OUT/fetch_$TAG.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_$TAG -o $TAG -- python $ROOT/tools/traffic_probe.py > $OUT/write_$TAG.log 2>&1; echo "write rc=$?"
ls $OUT/fetch_$TAG $OUT/write_$TAG
