#!/bin/bash
# bench once per value of an environment variable (first: unset).  usage: env_vals.sh VAR "kernel substrings" v1 v2 ...
V=$1; KS=$2; shift; shift
one() { python bench.py --cpu-gops 0 --steps 6 --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); t=d['roofline']['all_kernels_ms_one_step']
ks='$KS'.split()
print('$1', d['value'], d['ms_per_step'], 'sum %.2f' % sum(t.values()), {k:v for k,v in t.items() if any(x in k for x in ks)})"; }
one unset; one unset
for x in "$@"; do export $V=$x; one "$V=$x"; one "$V=$x"; done
