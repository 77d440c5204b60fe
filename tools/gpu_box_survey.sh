#!/bin/bash
# What kind of box is this one?  One short bench run (no extras), its headline and its `box` telemetry in one line - called a few
# times over a round, the lines show how much of the run-to-run spread is the machine (clock, power, cap) and how much is not.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys,time; j=json.loads(sys.stdin.readline()); b=j['box']
pl=list(b.get('sysfs_under_pipelined_load',{}).values()); sl=list(b.get('sysfs_under_single_stream_load',{}).values())
cap=[v.get('power1_cap') for v in b.get('sysfs',{}).values()]
kt={r['kernel'][:14]: round(r['ms']*1e3,1) for r in j['kernel_table']}
print(json.dumps({'utc': time.strftime('%Y-%m-%d %H:%M:%S', time.gmtime()), 'pci': b.get('pci'), 'value': j['value'], 'blocks': j['blocks']['img_s'], 'single_stream': j['single_stream_img_s'],
  'forward_ms': j['forward_only_ms'], 'sclk_cold': b['sclk_cold']['sclk_mhz_mean'], 'sclk_pipelined': b['sclk_during_pipelined_steps']['sclk_mhz_mean'],
  'sclk_one_stream': b['sclk_during_single_stream_steps']['sclk_mhz_mean'], 'power_pipelined_w': pl[0].get('power_w') if pl else None,
  'power_one_stream_w': sl[0].get('power_w') if sl else None, 'power_cap_uw': cap[0] if cap else None, 'perf_level': (b.get('rocm_smi') or {}).get('Performance Level'),
  'chain_us': kt.get('block_s1chain6'), 'stem_us': kt.get('stem_h3_kernel'), 's4h_us': kt.get('s4h_kernel')}))" | tee -a gpurun_out/box_survey.txt
