#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r2c; mkdir -p $OUT
echo "== key parity (chain on)"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "stage_activations or real_images or odd_batch or batch_invariance or end_to_end_survivors or 320 or fallback" > $OUT/pytest_key.log 2>&1; echo "rc=$?"; tail -15 $OUT/pytest_key.log
echo "== trace chain (step 6)"; timeout 120 python tools/trace_waves.py 6 2>&1 | grep -v amdgpu.ids | tee $OUT/trace_6.txt
echo "== timing"
for rep in 1 2; do
  echo "-- chain (#$rep)"; timeout 200 python tools/scale_probe.py 256 2>&1 | grep -E "stage3|stage4.0|conv1x1_2|TOTAL" | cut -c1-60,96-140
  echo "-- pairs (#$rep)"; YFV2_S1CHAIN=0 timeout 200 python tools/scale_probe.py 256 2>&1 | grep -E "stage3|stage4.0|conv1x1_2|TOTAL" | cut -c1-60,96-140
done
echo "== pytest -m gpu (all)"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_all.log 2>&1; echo "rc=$?"; tail -8 $OUT/pytest_all.log
