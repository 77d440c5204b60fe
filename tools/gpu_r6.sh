#!/bin/bash
# Round 6 evidence call (one tree, one fingerprint): GPU parity tests (+ parity counts), bench line, rocprofv3 kernel stats, SQ counter pass, HBM
# traffic passes - all on ONE tree, summarised with the tree's source fingerprint so that bench.py can quote them.
# usage (from the repo root on the GPU box): bash tools/gpu_r6.sh TAG [quick]
TAG=${1:-r06a}
MODE=${2:-full}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python tools/srchash.py > $OUT/src_hash.txt
echo "== pytest -m gpu"
timeout 1100 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -25 $OUT/pytest_gpu.log
cp gpurun_out/parity_counts.json $OUT/parity_counts.json 2>/dev/null
if [ "$MODE" = "quick" ]; then
  echo "== bench (quick)"
  timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench.log | cut -c1-3000
  timeout 300 python tools/scale_probe.py 1 256 > $OUT/scale_probe.txt 2>&1; tail -22 $OUT/scale_probe.txt | cut -c1-150
  echo "== post: fused / two launches"
  timeout 120 python tools/nms_probe.py 2>&1 | grep -v amdgpu.ids; YFV2_POSTFUSE=0 timeout 120 python tools/nms_probe.py 2>&1 | grep -v amdgpu.ids
  exit 0
fi
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
echo "== rocprofv3 kernel stats"   # --pipeline 1: one stream, so that a launch's duration is its own (overlapped launches share the machine)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o $TAG -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --pipeline 1 --blocks 2 > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"
find $OUT/prof -name "*kernel_stats*" | head -1 | while read f; do cp "$f" $OUT/kernel_stats.csv; head -30 "$f" | cut -c1-150; done
echo "== pmc SQ pass"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc1 -o $TAG -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --profile-iters 1 --pipeline 1 --blocks 1 --spinup-seconds 0.3 > $OUT/pmc1.log 2>&1; echo "pmc1 rc=$?"
python $ROOT/tools/pmc_table.py $OUT/pmc1 $OUT/pmc.json > $OUT/pmc_sq_summary.txt; head -40 $OUT/pmc_sq_summary.txt | cut -c1-140
echo "== traffic passes"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o $TAG -- python $ROOT/tools/traffic_probe.py > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o $TAG -- python $ROOT/tools/traffic_probe.py > $OUT/write.log 2>&1; echo "write rc=$?"
python $ROOT/tools/traffic_summary.py $OUT/fetch $OUT/write 380633088 > $OUT/traffic.json; python -c "
import json; j=json.load(open('$OUT/traffic.json')); print(j.get('calibration'))
for k,v in j['kernels'].items():
    if 'at::' not in k and 'rocclr' not in k: print('%-70s %8.1f MB' % (k[:70], v['total_bytes']/1e6), v.get('per_launch_total_bytes_in_dispatch_order',''))"
# the bench line LAST, with the summaries of this very tree in profiles/ so that it can quote them
cd $ROOT
cp $OUT/traffic.json profiles/${TAG}_traffic.json; cp $OUT/pmc.json profiles/${TAG}_pmc.json
echo "== bench"
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench.log | cut -c1-1500
echo "== evidence belongs to this tree"
YFV2_STRICT_EVIDENCE=1 timeout 300 python -m pytest tests/test_abi_and_host.py -q -p no:cacheprovider -k "committed_counter_profiles" 2>&1 | tail -2
echo "== scale probe"
timeout 300 python tools/scale_probe.py 1 32 256 > $OUT/scale_probe.txt 2>&1; echo "probe rc=$?"; tail -30 $OUT/scale_probe.txt | cut -c1-150
rm -rf $OUT/prof/*/*.db $OUT/fetch $OUT/write 2>/dev/null; du -sh $OUT
echo "== lanes probe"
timeout 300 python tools/lanes_probe.py 1 2 3 > $OUT/lanes_probe.txt 2>&1; tail -8 $OUT/lanes_probe.txt
echo "== post phases"
(timeout 120 python tools/trace_post.py random 0.3; timeout 120 python tools/trace_post.py coco 0.3) 2>&1 | grep -v amdgpu.ids > $OUT/post_phases.txt; cat $OUT/post_phases.txt
echo "== training iteration"
timeout 200 python tools/train_probe.py 8 64 2>&1 | grep "B=" > $OUT/train_probe.txt; cat $OUT/train_probe.txt
echo "== energy per launch"
timeout 200 python tools/power_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/power_probe.txt; cat $OUT/power_probe.txt | cut -c1-170
if [ -x tools/ubench/stem_pattern ]; then   # (built by hand from tools/ubench/stem_pattern.hip: the two-launch plan's access-pattern yardstick)
  echo "== stem access pattern"
  timeout 60 tools/ubench/stem_pattern > $OUT/stem_pattern.txt 2>&1; tail -9 $OUT/stem_pattern.txt
fi
