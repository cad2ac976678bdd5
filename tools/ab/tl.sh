#!/bin/bash
# batch-level timeline of the headline loop without a profiler (DSV1_TIMELINE=1: event marks on the pipeline's own streams)
# usage (through gpurun): tools/ab/tl.sh <tag> [env settings...]
REPO=${GRAFT_REPO_ROOT:-$PWD}; TAG=$1; shift
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO && env DSV1_TIMELINE=1 DSV1_HOST_PROF=1 "$@" python3 bench.py --cpu-gops 0 --steps 6 --no-extras --prof-kernel none > $OUT/${TAG}_tl_bench.json 2> $OUT/${TAG}_tl.txt
python3 tools/tl_show.py $OUT/${TAG}_tl.txt | tail -80
python3 -c "
import json; d=json.loads(open('$OUT/${TAG}_tl_bench.json').read().strip().splitlines()[-1]); print('value', d['value'], 'ms_per_step', d['ms_per_step'])"
