#!/bin/bash
# per-picture kernel times (alone on the chip) at small and large batches: does a frame step whose producer -> consumer data (prediction, symbols)
# fits the 256 MB Infinity Cache run its inverse kernels faster per picture?
for g in 16 32 64 320; do
  python3 bench.py --cpu-gops 0 --steps 6 --gops $g --no-extras 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); t=d['roofline']['all_kernels_ms_one_step']; g=$g
ks=['void k_inv_p_tile<true>','k_inv_patch_c','void k_fwd_mc_fast<0>','void k_fwd_mc_fast<1>','k_unpack','void k_hme_level<true>']
print('gops %4d  step %7.3f ms  per 320 GOPs: ' % (g, d['ms_per_step']) + '  '.join('%s %.2f' % (k.replace('void ','')[:16], t.get(k,0)*320.0/g) for k in ks))"
done
