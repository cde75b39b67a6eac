#!/bin/bash
# Build an experimental copy of libdvd_hip.so with extra -D flags for the warp+loss kernel:
#   tools/build_variant.sh <name> [-DDVD_WARP_PIN=0 ...]   ->  dvd_hip/lib/variants/libdvd_hip_<name>.so
# Select it at run time with DVD_HIP_LIB=<path> (dvd_hip/_lib.py).  Experiments only.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/dynamic-video-depth_amd/dvd_hip
OUT=$PKG/lib/variants
mkdir -p $OUT/obj_$NAME
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$ROOT/include -I$PKG/csrc"
hipcc $COMMON -ffp-contract=off "$@" -c $PKG/csrc/warp_loss.hip -o $OUT/obj_$NAME/warp_loss.o
hipcc --offload-arch=gfx950 -shared -fPIC $PKG/lib/core.o $PKG/lib/unproject.o $OUT/obj_$NAME/warp_loss.o \
      $PKG/lib/sf_mlp.o $PKG/lib/elementwise.o $PKG/lib/gconv.o $PKG/lib/gconv32.o $PKG/lib/surfaces.o $PKG/lib/upsample.o $PKG/lib/bnrelu.o -o $OUT/libdvd_hip_$NAME.so
echo $OUT/libdvd_hip_$NAME.so
