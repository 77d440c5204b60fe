#!/bin/bash
# One gpurun call: GPU parity tests, a short bench and a rocprofv3 kernel-trace summary.
# usage (from the repo root on the GPU box): bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/box.txt; nproc >> $OUT/box.txt; lscpu | grep "Model name" >> $OUT/box.txt
echo "== pytest -m gpu" 
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"
tail -40 $OUT/pytest_gpu_$TAG.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke_$TAG.log
echo "== bench"
timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/bench_$TAG.log 2>&1; echo "bench rc=$?"; tail -2 $OUT/bench_$TAG.log
echo "== scale probe"
timeout 300 python tools/scale_probe.py > $OUT/scale_$TAG.log 2>&1; echo "probe rc=$?"
echo "== resize probe"
timeout 120 python tools/resize_probe.py > $OUT/resize_$TAG.log 2>&1; echo "resize rc=$?"; tail -4 $OUT/resize_$TAG.log
echo "== rocprofv3"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o $TAG -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
ls $OUT/prof_$TAG 2>/dev/null | head; find $OUT/prof_$TAG -name "*kernel_stats*" | head -2 | while read f; do head -25 "$f"; done
