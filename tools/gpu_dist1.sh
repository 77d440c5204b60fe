#!/bin/bash
# bench.py under torch.distributed.run with ONE rank (RCCL path, forced gather) next to the plain run, same box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for rep in 1 2; do
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$rep bench.py --gpus 1 --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('torchrun x1:', d['value'], d['ms_per_step'], d['forward_only_ms'])"
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('plain      :', d['value'], d['ms_per_step'], d['forward_only_ms'])"
done
