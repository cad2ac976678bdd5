#!/bin/bash
# per-kernel exclusive times per GOP at several batch sizes (where does a larger batch stop paying?)
for g in $1; do
python bench.py --cpu-gops 4 --steps 4 --gops $g --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); t=d['roofline']['all_kernels_ms_one_step']; g=$g
print(g, d['value'], d['ms_per_step'], 'bit_exact', d['bit_exact_vs_cpu'], 'us/GOP:', {k.replace('void ',''):round(1000*v/g,2) for k,v in t.items() if v/g*1000>0.5})"
done
