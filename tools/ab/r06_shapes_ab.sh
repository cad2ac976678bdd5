#!/bin/bash
# same-box A/B of library variants (tools/ab/variant.sh) on the SMALL shapes (tools/bench_shape.py): usage r06_shapes_ab.sh base v1 base v1
REPO=${GRAFT_REPO_ROOT:-$PWD}
for v in "$@"; do
  if [ "$v" = base ]; then unset DSV1_SO; else export DSV1_SO=$REPO/digital-subband-video-1_amd/variants/$v/libdsv1_mi355x.so; fi
  echo "== $v"
  python3 $REPO/tools/bench_shape.py 3840 2160 2 16 12 85 1 0 16 2>/dev/null | tail -1
  python3 $REPO/tools/bench_shape.py 3840 2160 2 8 12 85 1 0 24 2>/dev/null | tail -1
  python3 $REPO/tools/bench_shape.py 1920 1080 2 4 12 85 1 0 60 2>/dev/null | tail -1
  python3 $REPO/tools/bench_shape.py 1920 1080 2 64 0 85 1 0 20 2>/dev/null | tail -1
done
