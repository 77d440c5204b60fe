cd $GRAFT_REPO_ROOT
echo "== quirky"; timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "pruned_and_collapsed" 2>&1 | tail -12
echo "== train tests"; timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q --tb=line -p no:cacheprovider 2>&1 | tail -3
timeout 200 python tools/train_probe.py 8 64 2>&1 | grep "B="
echo "== large batch other plans / sizes"
for cfg in "YFV2_BF6=0 6144 fp32 352 352" "YFV2_FUSED=0 1024 fp32 352 352" "X=1 2304 fp32 512 512" "X=1 3072 fp32 640 384" "X=1 6144 uint8 288 384" "YFV2_POSTFUSE=0 6144 fp32 352 352"; do
  set -- $cfg; echo "-- $cfg"; env $1 timeout 300 python tests/gpu_cases/large_batch.py $2 $3 $4 $5 2>&1 | grep -v amdgpu.ids | tail -3
done
