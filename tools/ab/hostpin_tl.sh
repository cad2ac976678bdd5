#!/bin/bash
# the host-fed loop (frames in pinned host memory, one upload per step) under the in-library timeline: does the upload run back to back,
# does coding overlap it?  usage (through gpurun): tools/ab/hostpin_tl.sh <tag> [env settings...]
cd ${GRAFT_REPO_ROOT:-$PWD}; TAG=$1; shift
env DSV1_TIMELINE=1 "$@" python3 bench.py --cpu-gops 0 --steps 4 --no-extras --prof-kernel none --input pinned > gpurun_out/${TAG}_pinned.json 2> gpurun_out/${TAG}_pinned_tl.txt
python3 tools/tl_show.py gpurun_out/${TAG}_pinned_tl.txt | cut -c1-420 | tee gpurun_out/${TAG}_pinned_tl_summary.txt
python3 -c "
import json; d=json.loads(open('gpurun_out/${TAG}_pinned.json').read().strip().splitlines()[-1]); print('$TAG $*: value', d['value'], 'ms_per_step', d['ms_per_step'])"
