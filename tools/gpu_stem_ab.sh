cd $GRAFT_REPO_ROOT
echo "== parity"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "stage_activations or real_images or seeded_rand or odd_batch or 320 or 288x384 or small_and_strip or uint8 or u8" 2>&1 | tail -6 | cut -c1-250
for rep in 1 2; do
echo "-- current #$rep"; timeout 200 python tools/scale_probe.py 256 2>&1 | grep -i "stem\|total\|forward" | cut -c1-60,96-140; timeout 200 python tools/u8_probe.py 2>&1 | grep -v amdgpu
echo "-- prev #$rep"; YFV2_LIB=$GRAFT_REPO_ROOT/yolo_fastestv2_amd/libyfv2_prev.so timeout 200 python tools/scale_probe.py 256 2>&1 | grep -i "stem\|total\|forward" | cut -c1-60,96-140; YFV2_LIB=$GRAFT_REPO_ROOT/yolo_fastestv2_amd/libyfv2_prev.so timeout 200 python tools/u8_probe.py 2>&1 | grep -v amdgpu
done
