#!/bin/bash
# Same-box A/B/C/.. of several environment configurations (plan switches, YFV2_VARIANT bits) of ONE built library:
#   bash tools/gpu_ab_configs.sh "PATTERN" "CFG0" "CFG1" ...        e.g.  "TOTAL|towers" "YFV2_TPAIR=0" "" "YFV2_VARIANT=11"
# ("" = the defaults).  Key parity tests run under every configuration first; then the per-launch probe (B = 256) for each,
# twice, alternating, so that box noise shows (NOPARITY=1: probes only).  Output under gpurun_out/ab_configs/.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
PAT=${1:-"TOTAL"}; shift
OUT=$ROOT/gpurun_out/ab_configs; mkdir -p $OUT
KEY="stage_activations or real_images or seeded_rand or odd_batch or more_images or batch_invariance or end_to_end_survivors or 320 or noise_floor or 288x384"
i=0
for cfg in "$@"; do
  [ -n "$NOPARITY" ] && break
  echo "== key parity under [$cfg]"
  env $cfg timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "$KEY" > $OUT/pytest_$i.log 2>&1; echo "rc=$?"; tail -4 $OUT/pytest_$i.log
  i=$((i+1))
done
for rep in 1 2; do
  i=0
  for cfg in "$@"; do
    echo "-- [$cfg] (#$rep)"; env $cfg timeout 200 python tools/scale_probe.py 256 > $OUT/probe_${i}_$rep.txt 2>&1; grep -E "$PAT" $OUT/probe_${i}_$rep.txt | cut -c1-62,96-118
    i=$((i+1))
  done
done
