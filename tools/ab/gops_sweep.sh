#!/bin/bash
# headline throughput against closed GOPs per step (same box): usage gops_sweep.sh "160 256 320" [reps]
for r in $(seq 1 ${2:-2}); do
for g in ${1:-160 224 320 448}; do
  timeout 300 python bench.py --cpu-gops 0 --steps 6 --gops $g --no-extras 2>/tmp/e.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$g', d['value'], d['ms_per_step'])" || tail -3 /tmp/e.txt
done
done
