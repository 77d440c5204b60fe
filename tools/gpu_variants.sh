#!/bin/bash
# same-box A/B/C of several built libraries: per-launch probe at B = 256, twice, alternating.
# usage: bash tools/gpu_variants.sh "grep pattern" lib1.so lib2.so ...   (paths relative to yolo_fastestv2_amd/)
PAT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for rep in 1 2; do
  for L in "$@"; do
    echo "-- $L (#$rep)"; YFV2_LIB=$ROOT/yolo_fastestv2_amd/$L timeout 200 python tools/scale_probe.py 256 2>&1 | grep -i "$PAT" | cut -c1-60,96-140
  done
done
