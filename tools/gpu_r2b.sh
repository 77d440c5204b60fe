#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r2b; mkdir -p $OUT
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest_all.log 2>&1; echo "rc=$?"; tail -5 $OUT/pytest_all.log
for st in 7 9 22 23 18 13; do
  echo "== trace step $st"; timeout 120 python tools/trace_waves.py $st 2>&1 | grep -v amdgpu.ids | tee $OUT/trace_$st.txt
done
echo "== s1w 512 vs 1024 threads (S1X2=0)"
for rep in 1 2; do
  YFV2_S1X2=0 timeout 200 python tools/scale_probe.py 256 2>&1 | grep -E "stage3|TOTAL" | cut -c1-50,96-140
  YFV2_S1X2=0 YFV2_S1W_T=1024 timeout 200 python tools/scale_probe.py 256 2>&1 | grep -E "stage3|TOTAL" | cut -c1-50,96-140
done
