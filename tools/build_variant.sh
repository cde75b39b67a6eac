#!/bin/bash
# Build an experimental copy of libdvd_hip.so with extra -D flags for ONE translation unit (default: the warp+loss kernel):
#   tools/build_variant.sh <name> [-DDVD_WARP_PINHOLE=0 ...]           ->  dvd_hip/lib/variants/libdvd_hip_<name>.so
#   UNIT=xconv tools/build_variant.sh <name> [-D...]
# Select it at run time with DVD_HIP_LIB=<path> (dvd_hip/_lib.py).  A/B experiments only; the product library is built by
# dvd_hip/build.py.
set -e
NAME=$1; shift
UNIT=${UNIT:-warp_loss}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/dynamic-video-depth_amd/dvd_hip
OUT=$PKG/lib/variants
mkdir -p $OUT/obj_$NAME
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$ROOT/include -I$PKG/csrc"
EXTRA=""
case $UNIT in warp_loss|warp_strip|unproject|elementwise|surfaces|upsample|consistency) EXTRA="-ffp-contract=off";; esac
SRC=${SRC:-$PKG/csrc/$UNIT.hip}          # SRC=<file>: another revision of the unit (git show <rev>:<path> > file)
hipcc $COMMON $EXTRA "$@" -c $SRC -o $OUT/obj_$NAME/$UNIT.o
OBJS=""
for o in $PKG/lib/*.o; do
  [ "$(basename $o)" = "$UNIT.o" ] && continue
  OBJS="$OBJS $o"
done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $OUT/obj_$NAME/$UNIT.o -o $OUT/libdvd_hip_$NAME.so
echo $OUT/libdvd_hip_$NAME.so
