#!/bin/bash
# same-box A/B of two builds: yolo_fastestv2_amd/libyfv2_prev.so (YFV2_LIB) against the current library - fingerprints (bit identity) and per-launch times,
# alternating twice; then stamps of the current build.   usage: bash tools/gpu_r6_ablib.sh "PATTERN" ["STEP NAME" ...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
PAT=${1:-towers}; shift
OUT=$ROOT/gpurun_out/r6_ablib; mkdir -p $OUT; rm -f $OUT/stamps.txt
for rep in 1 2; do
  echo "== prev"; YFV2_LIB=$ROOT/yolo_fastestv2_amd/libyfv2_prev.so timeout 300 python tools/variant_check.py "$PAT" 2>&1 | grep -v amdgpu.ids | tee $OUT/prev_$rep.txt | grep -E "fingerprint|total"
  echo "== current"; timeout 300 python tools/variant_check.py "$PAT" 2>&1 | grep -v amdgpu.ids | tee $OUT/cur_$rep.txt | grep -E "fingerprint|total"
done
for st in "$@"; do
  for B in 256 1; do
    echo "== [$st] B=$B"; timeout 200 python tools/trace_waves.py "$st" $B 2>&1 | grep -v amdgpu.ids | tee -a $OUT/stamps.txt | sed -n 1,3p\;6,7p | cut -c1-200
  done
done
