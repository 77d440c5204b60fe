#!/bin/bash
# Round-2 first GPU call: (1) class-count probes, each in its own interpreter, (2) the GPU suite on the merged tree,
# (3) key parity tests with each new kernel switched off, (4) same-box timing of merged vs previous build and switches.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r2a; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== class counts (merged tree)"
for c in 1 2 3 5 6 9 20 93; do
  timeout 150 python tests/gpu_cases/class_counts.py $c > $OUT/cls_$c.out 2> $OUT/cls_$c.err; rc=$?
  echo "classes=$c rc=$rc last: $(grep '^\[class_counts\]\|PARITY' $OUT/cls_$c.out | tail -1)"
  if [ $rc -ne 0 ]; then tail -15 $OUT/cls_$c.err; fi
done
echo "== class 1, layer-by-layer plan"
YFV2_FUSED=0 timeout 150 python tests/gpu_cases/class_counts.py 1 > $OUT/cls_1_unfused.out 2> $OUT/cls_1_unfused.err; echo "rc=$? $(tail -1 $OUT/cls_1_unfused.out)"; tail -5 $OUT/cls_1_unfused.err
echo "== pytest -m gpu (merged tree, all new kernels on)"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest_all.log 2>&1; echo "rc=$?"; tail -30 $OUT/pytest_all.log
KEY="stage_activations or real_images or odd_batch or batch_invariance or end_to_end_survivors or 320"
for sw in YFV2_S1X2 YFV2_S1W YFV2_DWPW; do
  echo "== key parity with $sw=0"
  env $sw=0 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line -p no:cacheprovider -k "$KEY" > $OUT/pytest_$sw.log 2>&1; echo "rc=$?"; tail -6 $OUT/pytest_$sw.log
done
echo "== key parity with all three off"
YFV2_S1X2=0 YFV2_S1W=0 YFV2_DWPW=0 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line -p no:cacheprovider -k "$KEY" > $OUT/pytest_alloff.log 2>&1; echo "rc=$?"; tail -6 $OUT/pytest_alloff.log
echo "== timing"
for rep in 1 2; do
  echo "-- merged (#$rep)"; timeout 200 python tools/scale_probe.py 256 > $OUT/probe_merged_$rep.txt 2>&1; tail -40 $OUT/probe_merged_$rep.txt | cut -c1-60,96-140
  echo "-- prev (#$rep)"; YFV2_LIB=$ROOT/yolo_fastestv2_amd/libyfv2_prev.so timeout 200 python tools/scale_probe.py 256 > $OUT/probe_prev_$rep.txt 2>&1; tail -40 $OUT/probe_prev_$rep.txt | cut -c1-60,96-140
done
echo "-- merged S1X2=0"; YFV2_S1X2=0 timeout 200 python tools/scale_probe.py 256 > $OUT/probe_s1x2off.txt 2>&1; grep -E "stage3|TOTAL" $OUT/probe_s1x2off.txt | cut -c1-60,96-140
echo "-- merged S1X2=0 S1W=0"; YFV2_S1X2=0 YFV2_S1W=0 timeout 200 python tools/scale_probe.py 256 > $OUT/probe_s1woff.txt 2>&1; grep -E "stage3|TOTAL" $OUT/probe_s1woff.txt | cut -c1-60,96-140
echo "-- merged DWPW=0"; YFV2_DWPW=0 timeout 200 python tools/scale_probe.py 256 > $OUT/probe_dwpwoff.txt 2>&1; grep -E "stage4.0|TOTAL" $OUT/probe_dwpwoff.txt | cut -c1-60,96-140
echo "== bench (merged)"
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.log 2>&1; echo "rc=$?"; tail -1 $OUT/bench.log | cut -c1-600
