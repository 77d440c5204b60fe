#!/bin/bash
# Same-box A/B of one environment switch:  bash tools/gpu_ab_env.sh VAR [grep-pattern]   (VAR=0 against the default)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
VAR=${1:-YFV2_BF6}; PAT=${2:-"TOTAL"}
OUT=$ROOT/gpurun_out/ab_$VAR; mkdir -p $OUT
echo "== key parity (default build)"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "stage_activations or real_images or odd_batch or batch_invariance or end_to_end or 320 or class_counts or fallback or uint8" > $OUT/pytest_key.log 2>&1; echo "rc=$?"; tail -6 $OUT/pytest_key.log
for rep in 1 2; do
  echo "-- default (#$rep)"; timeout 200 python tools/scale_probe.py 256 2>&1 | grep -E "$PAT" | cut -c1-60,96-140
  echo "-- $VAR=0 (#$rep)"; env $VAR=0 timeout 200 python tools/scale_probe.py 256 2>&1 | grep -E "$PAT" | cut -c1-60,96-140
done
