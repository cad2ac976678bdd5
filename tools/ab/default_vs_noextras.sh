#!/bin/bash
# same box: the default bench line against the --no-extras A/B form (does anything around the timed region cost?)
one() { python bench.py "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
one --cpu-gops 0 --no-extras --steps 10
one
one --cpu-gops 0 --no-extras --steps 10
one --cpu-gops 0
