#!/bin/bash
# A variant build of the library for an A/B measurement on the GPU box:
#     tools/build_variant.sh <name> [patch ...] [-- EXTRA=<flags>]
# copies csrc/ + include/ to a scratch tree, applies the patches (git-style, paths relative to the repo root), builds, and leaves
# tools/experiments/variants/liblio_hip_<name>.so (git-ignored, travels with gpurun).  Select it with LIO_HIP_LIB=<that path>.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
T=/tmp/lio_variant_$name
rm -rf $T && mkdir -p $T/lidar-slam-detection_amd/python/lsd_amd
cp -r $R/include $T/include
cp -r $R/lidar-slam-detection_amd/csrc $T/lidar-slam-detection_amd/csrc
rm -f $T/lidar-slam-detection_amd/csrc/*.o
while [ $# -gt 0 ]; do
    if [ "$1" = "--" ]; then shift; break; fi
    (cd $T && patch -p1 -s < "$R/$1")
    shift
done
# (what follows `--` goes to make as it is: quote an EXTRA with several flags, EXTRA="-DA=1 -DB=2")
make -C $T/lidar-slam-detection_amd/csrc -j8 ../python/lsd_amd/liblio_hip.so "$@" > $T/build.log 2>&1 || { tail -30 $T/build.log; exit 1; }
mkdir -p $R/tools/experiments/variants
cp $T/lidar-slam-detection_amd/python/lsd_amd/liblio_hip.so $R/tools/experiments/variants/liblio_hip_$name.so
echo "built tools/experiments/variants/liblio_hip_$name.so"
