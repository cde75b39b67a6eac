#!/bin/bash
# A whole-library variant: every translation unit with the extra flags.
#   tools/build_all_variant.sh <name> -DDVD_STREAM_STORES=1     ->  dvd_hip/lib/variants/libdvd_hip_<name>.so   (DVD_HIP_LIB=...)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/dynamic-video-depth_amd/dvd_hip
OUT=$PKG/lib/variants/all_$NAME
mkdir -p $OUT
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$ROOT/include -I$PKG/csrc"
pids=""
for src in $PKG/csrc/*.hip; do
  u=$(basename $src .hip)
  EXTRA=""
  case $u in warp_loss|warp_strip|unproject|elementwise|surfaces|upsample|consistency) EXTRA="-ffp-contract=off";; esac
  hipcc $COMMON $EXTRA "$@" -c $src -o $OUT/$u.o &
  pids="$pids $!"
  if [ $(echo $pids | wc -w) -ge 6 ]; then wait $pids; pids=""; fi
done
wait $pids
hipcc --offload-arch=gfx950 -shared -fPIC $OUT/*.o -o $PKG/lib/variants/libdvd_hip_$NAME.so
echo $PKG/lib/variants/libdvd_hip_$NAME.so
