#!/bin/bash
# Two short PMC passes over a forward-only run (kernel-trace only), summarised per kernel.
# usage: bash tools/gpu_pmc_quick.sh TAG [ENV=VAL ...]
TAG=${1:-pmcq}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  env "${EXTRA[@]}" timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/${name}_$TAG -o $TAG -- python $ROOT/tools/fwd_only.py > $OUT/${name}_$TAG.log 2>&1; echo "$name rc=$?"
}
EXTRA=("$@"); [ ${#EXTRA[@]} -eq 0 ] && EXTRA=(NONE=1)
run pmcA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run pmcB SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM
python $ROOT/tools/pmc_summary.py $OUT/pmcA_$TAG $OUT/pmcB_$TAG | tee $OUT/pmc_summary_$TAG.txt
