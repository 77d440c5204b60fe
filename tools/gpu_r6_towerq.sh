#!/bin/bash
# quick: bit identity + time of the 22x22 towers (towerp_kernel) against towerh_kernel<.., 2, 4> (YFV2_VARIANT=256), stamps of both launches
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r6_towerq; mkdir -p $OUT; rm -f $OUT/stamps.txt
for v in 256 0 256 0; do
  echo "== YFV2_VARIANT=$v"; YFV2_VARIANT=$v timeout 300 python tools/variant_check.py "towers" 2>&1 | grep -v amdgpu.ids | tee $OUT/variant_$v.txt | grep -E "fingerprint|total"
done
for st in "half a" "half b"; do
  for B in 256 1; do
    echo "== [$st] B=$B"; timeout 200 python tools/trace_waves.py "$st" $B 2>&1 | grep -v amdgpu.ids | tee -a $OUT/stamps.txt | sed -n 1,3p\;6,7p | cut -c1-200
  done
done
