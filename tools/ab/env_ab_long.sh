#!/bin/bash
# same-box A/B of an environment switch on the timed region alone: bench.py --steps N --no-extras, R rounds, on / off interleaved
# usage: env_ab_long.sh VAR [steps=60] [rounds=3]
V=$1; N=${2:-60}; R=${3:-3}
one() { python3 bench.py --cpu-gops 0 --steps $N --no-extras --prof-kernel none | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], 'timed output', (d.get('bit_exact_timed_output') or {}).get('equal'))"; }
for r in $(seq $R); do one on; export $V=1; one off; unset $V; done
