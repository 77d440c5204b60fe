#!/bin/bash
# Round 6, first call: where the tower / stage-4 / chain launches spend their cycles on today's box (per-wave stamps of workgroup 0, B = 256 and B = 1)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r6_stamps; mkdir -p $OUT
for st in "half a" "half b" "towers 11x11" "stage4.0" "stage3.1"; do
  for B in 256 1; do
    echo "== [$st] B=$B"; timeout 200 python tools/trace_waves.py "$st" $B 2>&1 | grep -v amdgpu.ids | tee -a $OUT/stamps.txt
  done
done
timeout 200 python tools/scale_probe.py 1 256 2>&1 | grep -v amdgpu.ids | cut -c1-62,96-130 | tee $OUT/scale.txt
