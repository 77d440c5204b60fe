#!/bin/bash
# Same-box A/B of env-switch variants: parity subset for each spec, then the B=256 per-launch probe,
# alternating twice.  usage: tools/gpu_ab.sh "<grep pattern>" SPEC1 SPEC2 ...   (SPEC = VAR=VALUE or NONE=1)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
PAT=$1; shift
for spec in "$@"; do
  echo "== parity $spec"
  env $spec timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "stage_activations or real_images or random_weights or batch_invariance or end_to_end_survivors or odd_batch or 320" 2>&1 | tail -4
done
for rep in 1 2; do
  for spec in "$@"; do
    echo "== probe $spec (#$rep)"; env $spec timeout 200 python tools/scale_probe.py 256 2>&1 | grep -E "$PAT" | cut -c1-40,96-140
  done
done
