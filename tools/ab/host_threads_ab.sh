#!/bin/bash
# headline throughput against host worker threads and batch size (is the cliff beyond 400 GOPs per step the host's?)
for g in 320 448; do for t in "" 24 48; do
  env ${t:+DSV1_HOST_THREADS=$t} python3 bench.py --cpu-gops 0 --steps 6 --gops $g --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('gops $g threads ${t:-default}', d['value'], d['ms_per_step'])"
done; done
