#!/bin/bash
# round 4, second call: all GPU tests on the tree with lanes + range guard; lanes probe; per-launch probe
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/r04b
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r04b/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r04b/pytest_gpu.log
cp gpurun_out/parity_counts.json gpurun_out/r04b/ 2>/dev/null
echo "== lanes probe"
timeout 400 python tools/lanes_probe.py 1 2 3 4 2>&1 | tee gpurun_out/r04b/lanes_probe.txt
echo "== scale probe"
timeout 300 python tools/scale_probe.py 256 2>&1 | tail -25 | cut -c1-150 | tee gpurun_out/r04b/scale_probe.txt
