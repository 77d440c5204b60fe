#!/bin/bash
# Same-box attribution of round 5's kernel changes under bench.py's own loops: the round-4 kernels (YFV2_FRONT=0: stem + stage2.0 as two
# launches; YFV2_VARIANT=96: s3h_kernel, stage4.0 as two bands x two roles) against the defaults, alternating twice.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for cfg in "YFV2_FRONT=0 YFV2_VARIANT=96" "YFV2_FRONT=1 YFV2_VARIANT=0"; do
    echo "== $cfg (#$rep)"
    env $cfg timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.0f  blocks %s  single-stream %.0f  forward-only %.4f ms (pipelined %.4f)  from uint8 %.4f ms  launches %d  sum %.4f ms' % (d['value'], [round(v/1e3,1) for v in d['blocks']['img_s']], d['single_stream_img_s'], d['forward_only_ms'], d['forward_only_pipelined_ms'], d['forward_from_uint8_hwc_ms'], d['forward_launches'], d['forward_sum_of_launch_ms']))
print('   pipelined: %s W, %.0f MHz; one stream: %s W, %.0f MHz' % (list(d['box'].get('sysfs_under_pipelined_load',{}).values())[0].get('power_w'), d['box']['sclk_during_pipelined_steps']['sclk_mhz_mean'], list(d['box'].get('sysfs_under_single_stream_load',{}).values())[0].get('power_w'), d['box']['sclk_during_single_stream_steps']['sclk_mhz_mean']))"
  done
done
