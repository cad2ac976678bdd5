#!/bin/bash
# same-box A/B of CU-masked stream settings (DSV1_CU_LOAD / _ANALYSIS / _CODE / _CODE2): each argument is one setting,
# e.g.  cumask_ab.sh "DSV1_CU_LOAD=0-31" "DSV1_CU_LOAD=0-47 DSV1_CU_CODE=48-255";  the baseline runs first and last
one() { python3 bench.py --cpu-gops 4 --steps 6 --no-extras 2>/tmp/cumask_err.txt | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); t=d['roofline']['all_kernels_ms_one_step']
print('%-60s' % '$1', d['value'], d['ms_per_step'], 'bit_exact', d['bit_exact_vs_cpu'], d.get('bit_exact_timed_output',{}).get('equal'))"; grep -h "stream on" /tmp/cumask_err.txt | sort | uniq -c | head -4; }
for s in "" "$@" ""; do env $s bash -c "$(declare -f one); one '${s:-baseline}'"; done
