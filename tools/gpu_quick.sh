#!/bin/bash
# Fast A/B loop on the GPU box (~40 s): key parity tests, then the per-launch probe at B=256
# with and without the environment switch given as $1 (e.g. YFV2_S1_48_HALF=1).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
SW=${1:-}
echo "== parity ($SW)"
env $SW timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "stage_activations or real_images or random_weights or batch_invariance or end_to_end_survivors" 2>&1 | tail -6
echo "== probe baseline"; timeout 200 python tools/scale_probe.py 256 2>&1 | grep -E "stem|stage2.0|stage3.0|stage3.1|TOTAL"
echo "== probe $SW"; env $SW timeout 200 python tools/scale_probe.py 256 2>&1 | grep -E "stem|stage2.0|stage3.0|stage3.1|TOTAL"
