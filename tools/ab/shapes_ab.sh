#!/bin/bash
# same-box A/B of an environment switch on the small shapes (config 4 at one GPU's share of 8 GOPs, config 5: 4K 4:4:4 ABR) and the headline
# usage (through gpurun): tools/ab/shapes_ab.sh VAR
V=$1
for r in 1 2; do
  for s in "" "$V=1"; do
    echo "== ${s:-default}"
    env $s python3 tools/bench_shape.py 3840 2160 2 8 12 85 1 0 12 2>/dev/null | tail -1
    env $s python3 tools/bench_shape.py 3840 2160 0 2 30 85 0 20000 6 2>/dev/null | tail -1
  done
done
