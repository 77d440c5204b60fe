#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r2d; mkdir -p $OUT
echo "== loss tests"
timeout 600 python -m pytest tests/test_loss.py tests/test_dropin.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_loss.log 2>&1; echo "rc=$?"; tail -25 $OUT/pytest_loss.log
echo "== small-batch forward latency"
timeout 300 python tools/scale_probe.py 1 8 32 256 2>&1 | grep -v amdgpu > $OUT/scale_small.txt; tail -1 $OUT/scale_small.txt; cut -c1-60,96-160 $OUT/scale_small.txt | head -30
