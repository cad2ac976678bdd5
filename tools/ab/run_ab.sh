#!/bin/bash
# A/B one kernel source on the same GPU box: bench twice with the tree as shipped, then swap in the alternative
# source, rebuild, bench twice.  usage: run_ab.sh <file under csrc> <alternative source> <kernel substring>
F=$1; ALT=$2; K=$3
one() { python bench.py --cpu-gops 0 --steps 6 --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['roofline']['all_kernels_ms_one_step']; print('$1', d['value'], d['ms_per_step'], {k:v for k,v in t.items() if '$K' in k})"; }
one new; one new
cp digital-subband-video-1_amd/csrc/$F /tmp/keep.hip; cp $ALT digital-subband-video-1_amd/csrc/$F
make -C digital-subband-video-1_amd/csrc -j8 > /dev/null 2>&1
one alt; one alt
cp /tmp/keep.hip digital-subband-video-1_amd/csrc/$F
make -C digital-subband-video-1_amd/csrc -j8 > /dev/null 2>&1
