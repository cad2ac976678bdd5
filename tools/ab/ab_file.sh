#!/bin/bash
# same-box A/B of one source file: bench with the tree as built, then with <file> replaced by <alt>, then restored
# usage: ab_file.sh <kernel substrings> <file> <alt>
KS=$1; F=$2; ALT=$3
one() { python bench.py --cpu-gops 0 --steps 6 --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); t=d['roofline']['all_kernels_ms_one_step']
ks='$KS'.split()
print('$1', d['value'], d['ms_per_step'], 'sum %.2f' % sum(t.values()), {k:v for k,v in t.items() if any(x in k for x in ks)})"; }
one new; one new
cp $F /tmp/ab_keep
# the source and the shipped build come back whatever happens in between (a failed alt build, an interrupt)
restore() { cp /tmp/ab_keep $F; touch $F; make -C digital-subband-video-1_amd/csrc -j8 > /dev/null 2>&1; }
trap restore EXIT
cp $ALT $F
make -C digital-subband-video-1_amd/csrc -j8 > /dev/null 2>&1
one alt; one alt
cp /tmp/ab_keep $F
make -C digital-subband-video-1_amd/csrc -j8 > /dev/null 2>&1
one new
