#!/bin/bash
# Round 6: towerp_kernel (channel-pair depthwise units) against towerh_kernel<.., 2, 4> (YFV2_VARIANT=256): bit identity, parity, time, stamps
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r6_towerp; mkdir -p $OUT
for v in 256 0 256 0; do
  echo "== YFV2_VARIANT=$v"; YFV2_VARIANT=$v timeout 300 python tools/variant_check.py "towers" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/variant_$v.txt
done
echo "== key parity (new kernel)"
KEY="stage_activations or real_images or seeded_rand or odd_batch or more_images or batch_invariance or end_to_end_survivors or 320 or noise_floor or 288x384 or class or sizes"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "$KEY" > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -5 $OUT/pytest.log
for st in "half a" "half b"; do
  for B in 256 1; do
    echo "== [$st] B=$B"; timeout 200 python tools/trace_waves.py "$st" $B 2>&1 | grep -v amdgpu.ids | tee -a $OUT/stamps.txt
  done
done
