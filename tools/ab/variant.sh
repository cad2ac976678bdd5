#!/bin/bash
# Build an A/B variant of the library next to the shipped one, here (cross-compiled), so that a GPU call only runs it:
#   tools/ab/variant.sh <name> "<objects to recompile, e.g. k_sbt>" "<extra hipcc flags, e.g. -DAB_FOO>"
# -> digital-subband-video-1_amd/variants/<name>/libdsv1_mi355x.so (git-ignored, travels with gpurun); select with DSV1_SO=<path>.
# The shipped build directory is copied, so only the named objects are recompiled; the shipped library is never touched.
set -e
REPO=$(cd "$(dirname "$0")/../.." && pwd)
N=$1; OBJS=$2; FLAGS=$3; CFL=$4          # $4: extra flags for the C session layer (objects host_dsv1_enc, host_dsv1_util, host_dsv1_dec)
SRC=$REPO/digital-subband-video-1_amd/csrc
V=$REPO/digital-subband-video-1_amd/variants/$N
mkdir -p "$V"
rm -rf "$V/build"; cp -r "$SRC/build" "$V/build"
for o in $OBJS; do rm -f "$V/build/$o.o"; done
make -C "$SRC" -j8 BUILD="$V/build" OUT="$V/libdsv1_mi355x.so" EXTRA="$FLAGS" EXTRA_C="$CFL" > "$V/build.log" 2>&1 || { tail -20 "$V/build.log"; exit 1; }
rm -rf "$V/build"
echo "$V/libdsv1_mi355x.so"
