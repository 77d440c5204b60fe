#!/bin/bash
# rocprofv3 kernel stats of a short bench run, top kernels only:  bash tools/gpu_prof_quick.sh TAG
TAG=${1:-q}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o $TAG -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
find $OUT/prof_$TAG -name "*kernel_stats*" | head -1 | while read f; do head -22 "$f" | cut -d, -f1-4,6,7 | cut -c1-160; done
