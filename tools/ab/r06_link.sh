#!/bin/bash
# round 6: how the headline step reacts to a slower link (DSV1_DEBUG_LINK_REPEAT=n: every large copy n times) -- same box, same binary
OUT=gpurun_out/${1:-r06_link}; mkdir -p $OUT
for n in 1 2 3 1; do
  DSV1_DEBUG_LINK_REPEAT=$n python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --cpu-gops 4 > $OUT/rep$n.json 2> $OUT/rep$n.err
  python3 - $OUT/rep$n.json $n <<'P'
import json,sys
p=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
sb=p["step_breakdown"]
print("repeat", sys.argv[2], p["value"], p["ms_per_step"], "idle", sb["device_idle_ms_per_batch"], "fetch", sb["fetch_host_ms_per_call"], "host", {k.split(" (")[0]: v for k, v in sb["host_ms_per_batch"].items()})
P
done
