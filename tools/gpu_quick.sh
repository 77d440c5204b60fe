#!/bin/bash
# Fast same-box A/B (~1 min): key parity tests on the current build, then the per-launch probe at
# B=256 for the current build and for a second build (default yolo_fastestv2_amd/libyfv2_prev.so,
# see tools/build_variant.sh), alternating twice to expose box noise.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OTHER=${OTHER:-$ROOT/yolo_fastestv2_amd/libyfv2_prev.so}
PAT=${1:-"stem|stage2.0|stage2.1|stage3.0|stage3.1|stage4.1|TOTAL"}
echo "== parity (current build)"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "stage_activations or real_images or odd_batch or batch_invariance or end_to_end_survivors or 320" 2>&1 | tail -4
for rep in 1 2; do
  echo "== probe current (#$rep)"; timeout 200 python tools/scale_probe.py 256 2>&1 | grep -E "$PAT" | cut -c1-40,96-140
  echo "== probe other   (#$rep)"; YFV2_LIB=$OTHER timeout 200 python tools/scale_probe.py 256 2>&1 | grep -E "$PAT" | cut -c1-40,96-140
done
