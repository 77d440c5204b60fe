#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r2f; mkdir -p $OUT
echo "== key parity"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "stage_activations or real_images or odd_batch or batch_invariance or end_to_end or 320 or class_counts or fallback or uint8" > $OUT/pytest_key.log 2>&1; echo "rc=$?"; tail -4 $OUT/pytest_key.log
PAT="stage3.1|stage4.0.main.pw1|conv1x1|TOTAL"
for rep in 1 2; do
  echo "-- current (#$rep)"; timeout 200 python tools/scale_probe.py 256 2>&1 | grep -E "$PAT" | cut -c1-60,96-140
  echo "-- prev = bf16x6 commit: split-on-the-fly pw, old chain phase 0 (#$rep)"; YFV2_LIB=$ROOT/yolo_fastestv2_amd/libyfv2_prev.so timeout 200 python tools/scale_probe.py 256 2>&1 | grep -E "$PAT" | cut -c1-60,96-140
done
echo "== chain trace (current)"; timeout 100 python tools/trace_waves.py 6 2>&1 | grep -v amdgpu
