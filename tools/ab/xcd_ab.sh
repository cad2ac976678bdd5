#!/bin/bash
# A/B of the XCD-ordered one-dimensional launches (k_inv_p_tile, k_fwd_mc_fast, k_inv_patch_c) against the hardware's
# round-robin order (DSV1_NO_XCD_ORDER=1), same box, same binary; then FETCH_SIZE / WRITE_SIZE per kernel both ways.
REPO=$PWD
OUT=$REPO/gpurun_out/${1:-xcd}
mkdir -p $OUT
one() { python3 bench.py --cpu-gops 0 --steps 6 --no-extras | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); t=d['roofline']['all_kernels_ms_one_step']
ks='k_inv_p_tile k_fwd_mc_fast k_inv_patch_c'.split()
print('$1', d['value'], d['ms_per_step'], 'sum %.2f' % sum(t.values()), {k:v for k,v in t.items() if any(x in k for x in ks)})"; }
( one xcd; DSV1_NO_XCD_ORDER=1 one plain; one xcd; DSV1_NO_XCD_ORDER=1 one plain ) 2>&1 | tee $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
PB="python3 $REPO/bench.py --cpu-gops 0 --steps 1 --warmup 1 --gops 160 --prof-kernel none --no-extras"
for v in xcd plain; do
  if [ $v = plain ]; then export DSV1_NO_XCD_ORDER=1; else unset DSV1_NO_XCD_ORDER; fi
  rm -rf /tmp/pf /tmp/pw
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- $PB > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- $PB > /dev/null 2>&1
  python3 $REPO/tools/pmc_summary.py "/tmp/pf/**/*counter_collection.csv" "/tmp/pw/**/*counter_collection.csv" > $OUT/pmc_hbm_$v.csv
  echo "== $v"; grep -E "k_inv_p_tile|k_fwd_mc_fast|k_inv_patch_c|kernel" $OUT/pmc_hbm_$v.csv
done
