#!/bin/bash
# PMC counter pass (own run, kernel-trace only - never combined with sys/hip traces).
# usage: bash tools/gpu_pmc.sh TAG
TAG=${1:-pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
echo "== scale probe"
timeout 300 python tools/scale_probe.py 16 64 256 > $OUT/scale_$TAG.log 2>&1; echo "probe rc=$?"; cat $OUT/scale_$TAG.log | cut -c1-160
cd /tmp && export TMPDIR=/tmp
echo "== pmc pass 1 (SQ)"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc1_$TAG -o $TAG -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-iters 1 > $OUT/pmc1_$TAG.log 2>&1; echo "pmc1 rc=$?"
echo "== pmc pass 2 (instruction mix)"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/pmc2_$TAG -o $TAG -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-iters 1 > $OUT/pmc2_$TAG.log 2>&1; echo "pmc2 rc=$?"
ls $OUT/pmc1_$TAG $OUT/pmc2_$TAG
tail -3 $OUT/pmc1_$TAG.log | cut -c1-300
