#!/bin/bash
# round 5 experiment: the post as one launch (default) against two launches (YFV2_POSTFUSE=0) under the bench's own loops, same box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/x1; mkdir -p $OUT
for rep in 1 2; do
for cfg in "" "YFV2_POSTFUSE=0"; do
  env $cfg timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_${rep}_${cfg:-default}.log 2>&1
  tail -1 $OUT/bench_${rep}_${cfg:-default}.log | python -c "
import json,sys; j=json.loads(sys.stdin.readline()); b=j['box']
print('[%s] value %.0f (blocks %s) single %.0f fwd_ms %.4f two_lanes %.0f | sclk busy %.0f pipelined %.0f single %.0f' % ('$cfg', j['value'], j['blocks']['img_s'], j['single_stream_img_s'], j['forward_only_ms'], j['single_call_two_lanes_img_s'], b['sclk_after_timed']['sclk_mhz_mean'], b['sclk_during_pipelined_steps']['sclk_mhz_mean'], b['sclk_during_single_stream_steps']['sclk_mhz_mean']))
print('   chain %.1f us stem %.1f us' % tuple(1e3*[r['ms'] for r in j['kernel_table'] if k in r['kernel']][0] for k in ('chain','stem')))"
done; done
