#!/bin/bash
# same-box A/B of prebuilt library variants (tools/ab/variant.sh): bench.py --no-extras once per variant, "base" = the shipped library
# usage (through gpurun, from the repo root): tools/ab/run_variants.sh "<kernel substrings>" base v1 v2 ... [base]
KS=$1; shift
REPO=${GRAFT_REPO_ROOT:-$PWD}
GOPS=${AB_GOPS:-320}
for v in "$@"; do
  if [ "$v" = base ]; then unset DSV1_SO; else export DSV1_SO=$REPO/digital-subband-video-1_amd/variants/$v/libdsv1_mi355x.so; fi
  python $REPO/bench.py --cpu-gops 0 --steps ${AB_STEPS:-4} --gops $GOPS --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); t=d['roofline']['all_kernels_ms_one_step']
ks='$KS'.split()
print('%-10s' % '$v', d['value'], d['ms_per_step'], 'sum %.2f' % sum(t.values()), {k.replace('void ',''):v for k,v in t.items() if any(x in k for x in ks)})"
done
