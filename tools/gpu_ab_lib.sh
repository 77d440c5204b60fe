#!/bin/bash
# Same-box A/B of two BUILDS: the tree's libyfv2.so against yolo_fastestv2_amd/libyfv2_prev.so (YFV2_LIB), key parity first
# usage: bash tools/gpu_ab_lib.sh [grep-pattern]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
PAT=${1:-"TOTAL"}
OUT=$ROOT/gpurun_out/ab_lib; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "stage_activations or real_images or odd_batch or batch_invariance or end_to_end or 320 or 288 or small_and_strip" > $OUT/pytest_key.log 2>&1; echo "rc=$?"; tail -2 $OUT/pytest_key.log
for rep in 1 2; do
  echo "-- new (#$rep)"; timeout 200 python tools/scale_probe.py 256 2>&1 | grep -E "$PAT" | cut -c1-60,96-140
  echo "-- prev (#$rep)"; YFV2_LIB=$ROOT/yolo_fastestv2_amd/libyfv2_prev.so timeout 200 python tools/scale_probe.py 256 2>&1 | grep -E "$PAT" | cut -c1-60,96-140
done
