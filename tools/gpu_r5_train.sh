#!/bin/bash
# Round 5, second half: the maximum-size parity case (tests/gpu_cases/large_batch.py) and a per-kernel picture of ONE training iteration
# (rocprofv3 --kernel-trace --stats over tools/train_probe.py 64).  usage (repo root, GPU box): bash tools/gpu_r5_train.sh TAG [notest]
TAG=${1:-r05t}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python tools/srchash.py > $OUT/src_hash.txt
if [ "$2" != "notest" ]; then
  echo "== large batch"
  timeout 700 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "beyond_4_gib" > $OUT/large_batch.log 2>&1; echo "rc=$?"; tail -15 $OUT/large_batch.log
fi
echo "== training tests"
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/train_tests.log 2>&1; echo "rc=$?"; tail -5 $OUT/train_tests.log
echo "== train probe"
timeout 200 python tools/train_probe.py 8 64 2>&1 | grep "B=" | tee $OUT/train_probe.txt
cd /tmp && export TMPDIR=/tmp
echo "== rocprofv3 over a training loop"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o $TAG -- python $ROOT/tools/train_probe.py 64 > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"
find $OUT/prof -name "*kernel_stats*" | head -1 | while read f; do cp "$f" $OUT/train_kernel_stats.csv; head -45 "$f" | cut -c1-170; done
rm -rf $OUT/prof
