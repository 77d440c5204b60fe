#!/bin/bash
# same-box A/B of the bench line's headline numbers: previous build (yolo_fastestv2_amd/libyfv2_prev.so through YFV2_LIB) against the current one,
# alternating twice.   usage: bash tools/gpu_r6_benchab.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for rep in 1 2; do
  for which in prev cur; do
    if [ $which = prev ]; then export YFV2_LIB=$ROOT/yolo_fastestv2_amd/libyfv2_prev.so; else unset YFV2_LIB; fi
    timeout 300 python bench.py --no-extras --no-cpu-baseline --blocks 9 2>/dev/null | tail -1 | python -c "
import json,sys
b=json.loads(sys.stdin.read())
print('$which', 'value %.1f k  single %.1f k  forward_sum %.4f ms  blocks %.1f-%.1f' % (b['value']/1e3, b['single_stream_img_s']/1e3, b.get('forward_sum_of_launch_ms',0), b['blocks']['img_s_min']/1e3, b['blocks']['img_s_max']/1e3))"
  done
done
