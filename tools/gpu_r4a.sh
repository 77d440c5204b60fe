#!/bin/bash
# round 4, first call: exception-word probe, the new noise-floor / ADVICE tests
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/r04a
echo "== excp probe"; timeout 60 tools/ubench/excp 2>&1 | tee gpurun_out/r04a/excp.txt
echo "== new tests"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_train_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "noise_floor or bench_regime or odd_batches or more_images or 320 or 288 or small_and_strip or uint8_entry or eval_after_training or train_bind" 2>&1 | tail -40 | tee gpurun_out/r04a/pytest.txt
cp gpurun_out/parity_counts.json gpurun_out/r04a/ 2>/dev/null
