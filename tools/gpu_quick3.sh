#!/bin/bash
# Round 3 quick call: key parity tests (-k pattern), then the per-launch probe at B = 256 (twice), optionally also for a
# second library (YFV2_LIB=...) for a same-box A/B.   usage: bash tools/gpu_quick3.sh TAG ["pytest -k pattern"] [other.so]
TAG=${1:-q}
PAT=${2:-"stage_activations or real_images or seeded_rand or odd_batch or batch_invariance or 320 or 288x384 or small_and_strip or fallback or uint8 or bench_regime or end_to_end"}
OTHER=$3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
echo "== key parity"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "$PAT" > $OUT/pytest_key.log 2>&1; echo "rc=$?"; tail -15 $OUT/pytest_key.log | cut -c1-300
for rep in 1 2; do
  echo "-- current (#$rep)"; timeout 200 python tools/scale_probe.py 256 2>&1 | grep -v amdgpu | cut -c1-60,96-140
  if [ -n "$OTHER" ]; then echo "-- other: $OTHER (#$rep)"; YFV2_LIB=$ROOT/$OTHER timeout 200 python tools/scale_probe.py 256 2>&1 | grep -v amdgpu | cut -c1-60,96-140; fi
done
