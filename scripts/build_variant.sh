#!/bin/bash
# experiment helper: build libmcorr with extra -D flags for ONE source into variants/<tag>/libmcorr.so
# usage: scripts/build_variant.sh <tag> <source.hip> <extra hipcc flags...>   (run on the CPU box)
set -e
tag=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
pkg=torch_motion_correction_amd
mkdir -p variants/$tag
base=$(basename $src .hip)
extra=""; [ $base = warp ] && extra="-fno-slp-vectorize"   # as torch_motion_correction_amd/_build.py does
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Iinclude -I$pkg/csrc $extra "$@" -c $pkg/csrc/$src -o variants/$tag/$base.o
objs=""
for o in plan_stats xc_fft xcg_fft_p0 xcg_fft_p1 xcg_fft_p2 xcg_fft_p3 field_post warp local_motion polyphase full_fft; do
  if [ $o = $base ]; then objs="$objs variants/$tag/$base.o"; else objs="$objs $pkg/build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o variants/$tag/libmcorr.so
echo built variants/$tag/libmcorr.so
