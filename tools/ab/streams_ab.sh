#!/bin/bash
# same-box A/B of the number of coding streams (and the packet fetch on the analysis stream) at the default batch size
for v in "" "DSV1_CODE_STREAMS=3" "DSV1_CODE_STREAMS=3 DSV1_FETCH_ON_ANALYSIS=1" "DSV1_CODE_STREAMS=1" ""; do
  env $v python3 bench.py --cpu-gops 0 --steps 6 --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', d['value'], d['ms_per_step'])"
done
