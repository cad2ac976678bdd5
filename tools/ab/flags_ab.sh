#!/bin/bash
# like run_ab_flags.sh with several kernel substrings and the bit-exact flag printed
# usage: flags_ab.sh "<kernel substrings>" "<flags A>" ["<flags B>" ...]
KS=$1; shift
one() { python bench.py --cpu-gops 0 --steps 6 --no-extras | python -c "
import sys,json
d=json.loads(sys.stdin.read()); t=d['roofline']['all_kernels_ms_one_step']
ks='$KS'.split()
print('$1', d['value'], d['ms_per_step'], 'sum %.2f' % sum(t.values()), {k:v for k,v in t.items() if any(x in k for x in ks)})"; }
one base; one base
for F in "$@"; do
  touch digital-subband-video-1_amd/csrc/*.hip
  make -C digital-subband-video-1_amd/csrc -j8 EXTRA="$F" > /dev/null 2>&1
  one "[$F]"; one "[$F]"
done
