#!/bin/bash
# The shader clock the hot kernels run at, measured inside the kernels (s_memtime / s_memrealtime, dsvg_dev.hpp
# DSVG_CLOCK_PROBE) while the bench workload runs un-profiled.  Rebuilds csrc with the probe, runs the headline workload,
# prints the "[clock probe]" lines the library writes when the context is destroyed, then restores the shipped build.
# usage (through gpurun, from the repo root): bash tools/ab/clock_probe.sh [gops]
GOPS=${1:-160}
# whatever happens (a failed probe build, an interrupt), the shipped build is restored: probe objects newer than their sources
# would otherwise be linked into libdsv1_mi355x.so by a later plain `make`
restore() { touch digital-subband-video-1_amd/csrc/*.hip; make -C digital-subband-video-1_amd/csrc -j8 > /dev/null 2>&1; }
trap restore EXIT
touch digital-subband-video-1_amd/csrc/*.hip
make -C digital-subband-video-1_amd/csrc -j8 EXTRA=-DDSVG_CLOCK_PROBE > /dev/null 2>&1 || { echo "probe build failed"; exit 1; }
for i in 1 2; do
  python bench.py --cpu-gops 0 --steps 8 --gops $GOPS --no-extras 2> /tmp/clk.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('probe build', d['value'], d['ms_per_step'])"
  grep "clock probe" /tmp/clk.err
done
