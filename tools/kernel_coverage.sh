#!/bin/bash
# Which kernels of libyfv2.so does the GPU suite ever launch?  (1) every __global__ symbol of every translation unit (device-only compile, symbol
# table), (2) rocprofv3 --kernel-trace over `pytest -m gpu` (child interpreters included), (3) the difference = kernels no test reaches.
# usage (GPU box): bash tools/kernel_coverage.sh      -> gpurun_out/cov/{all_kernels.txt,launched.txt,never_launched.txt}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/cov; rm -rf $OUT; mkdir -p $OUT/obj
cd /tmp && export TMPDIR=/tmp
for f in $ROOT/yolo_fastestv2_amd/csrc/*.hip; do
  u=$(basename $f .hip)
  extra=""; case $u in yfv2_post|yfv2_loss|yfv2_train) extra="-ffp-contract=off";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -std=c++20 -fPIC -Wno-everything $extra --cuda-device-only --no-gpu-bundle-output -c $f -o $OUT/obj/$u.o 2>/dev/null
  /opt/rocm/lib/llvm/bin/llvm-readelf -s --wide $OUT/obj/$u.o | awk '$4=="OBJECT" && $8 ~ /\.kd$/ {sub(/\.kd$/,"",$8); print $8}' | c++filt | sed 's/^void //; s/([^()]*)$//' | sort -u | sed "s/^/$u /"
done > $OUT/all_kernels.txt
cd $ROOT
timeout 1500 rocprofv3 --kernel-trace -d $OUT/prof -o cov_%pid% -- python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
tail -2 $OUT/pytest.log
python3 tools/kernel_coverage_report.py $OUT
rm -rf $OUT/obj $OUT/prof
