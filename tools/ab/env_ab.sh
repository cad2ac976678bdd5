#!/bin/bash
# A/B a run-time switch on the same GPU box: bench twice as built, twice with the environment variable set.
# usage: env_ab.sh <VAR> [kernel substring ...]
V=$1; shift
KS="$*"
one() { python bench.py --cpu-gops 0 --steps 6 --no-extras | python -c "
import sys,json
d=json.loads(sys.stdin.read()); t=d['roofline']['all_kernels_ms_one_step']
ks='$KS'.split()
print('$1', d['value'], d['ms_per_step'], d.get('bit_exact_vs_cpu'), 'sum %.2f' % sum(t.values()), {k:v for k,v in t.items() if any(x in k for x in ks)})"; }
one new; one new
export $V=1
one "$V"; one "$V"
