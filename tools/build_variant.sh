#!/bin/bash
# Build the library from a given git revision into yolo_fastestv2_amd/libyfv2_<tag>.so (for same-box A/B).
# usage: tools/build_variant.sh <git-rev> <tag>
set -e
REV=$1; TAG=$2; ROOT=$(cd $(dirname $0)/..; pwd); TMP=$(mktemp -d)
mkdir -p $TMP/include $TMP/yolo_fastestv2_amd/csrc
git -C $ROOT archive $REV include yolo_fastestv2_amd/csrc | tar -x -C $TMP
make -C $TMP/yolo_fastestv2_amd/csrc -j4 > /dev/null 2>&1
cp $TMP/yolo_fastestv2_amd/libyfv2.so $ROOT/yolo_fastestv2_amd/libyfv2_$TAG.so
rm -rf $TMP; ls -la $ROOT/yolo_fastestv2_amd/libyfv2_$TAG.so
