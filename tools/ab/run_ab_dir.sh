#!/bin/bash
# A/B the working tree against the csrc files kept under tools/ab/alt/ (same names) on the same GPU box:
# bench twice as is, swap the alt files in, rebuild, bench twice, restore.  usage: run_ab_dir.sh <kernel substring> [gops]
K=$1; G=${2:-96}
C=digital-subband-video-1_amd/csrc
one() { python bench.py --cpu-gops 0 --steps 20 --gops $G | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['roofline']['all_kernels_ms_one_step']; print('$1', d['value'], d['ms_per_step'], {k:v for k,v in t.items() if '$K' in k})"; }
one new; one new
mkdir -p /tmp/keep_csrc
for f in tools/ab/alt/*; do b=$(basename $f); cp $C/$b /tmp/keep_csrc/$b; cp $f $C/$b; done
make -C $C -j8 > /dev/null 2>&1
one alt; one alt
for f in tools/ab/alt/*; do b=$(basename $f); cp /tmp/keep_csrc/$b $C/$b; done
make -C $C -j8 > /dev/null 2>&1
one new
