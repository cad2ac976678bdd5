#!/bin/bash
# same-box A/B of an environment switch: bench.py --steps 6 --no-extras twice each way; usage: env_ab2.sh VAR "<kernel substrings>"
V=$1; export KS="$2"
one() { python3 bench.py --cpu-gops 4 --steps 6 --no-extras | python3 -c "
import sys,json,os
d=json.loads(sys.stdin.read()); t=d['roofline']['all_kernels_ms_one_step']
ks=os.environ['KS'].split()
print('$1', d['value'], d['ms_per_step'], 'bit_exact', d['bit_exact_vs_cpu'], 'sum %.2f' % sum(t.values()), {k:v for k,v in t.items() if any(x in k for x in ks)})"; }
one on; env $V=1 bash -c "$(declare -f one); one off"; one on; env $V=1 bash -c "$(declare -f one); one off"
