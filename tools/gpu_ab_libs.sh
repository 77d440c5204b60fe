#!/bin/bash
# Same-box A/B of several builds: the per-launch probe at B = 256 for the in-tree library and for every extra .so given
# (YFV2_LIB override), alternating, REPS times; only the lines matching PAT are shown.
# usage: bash tools/gpu_ab_libs.sh "PAT" REPS lib1.so [lib2.so ...]      (paths relative to the repo root)
PAT=${1:-"stem|TOTAL"}; REPS=${2:-2}; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for rep in $(seq 1 $REPS); do
  echo "-- in-tree (#$rep)"; timeout 200 python tools/scale_probe.py 256 2>&1 | grep -E "$PAT" | cut -c1-60,96-140
  for lib in "$@"; do echo "-- $lib (#$rep)"; YFV2_LIB=$ROOT/$lib timeout 200 python tools/scale_probe.py 256 2>&1 | grep -E "$PAT" | cut -c1-60,96-140; done
done
