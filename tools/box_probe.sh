#!/bin/bash
# What kind of box is this?  (the same tree measures 1.06 ms per step on most boxes and 1.31-1.39 ms on some)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
rocm-smi --showproductname --showclocks --showpower --showmaxpower --showperflevel --showmemuse 2>/dev/null | grep -v "^=\|^$" | head -30
rocm-smi --showcomputepartition --showmemorypartition 2>/dev/null | grep -v "^=\|^$" | head -8
python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print("CUs", p.multi_processor_count, "clock_rate_khz", getattr(p, "clock_rate", None), "mem GB", round(p.total_memory / 2**30, 1), "L2", p.L2_cache_size if hasattr(p, "L2_cache_size") else None)
PY
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('bench:', d['value'], d['ms_per_step'], d['forward_only_ms'])"
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | head -6
