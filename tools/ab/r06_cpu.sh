#!/bin/bash
# round 6: how the headline step reacts to a starved host (taskset to a few cores; the NUMA placement probe is off so that the mask stays)
OUT=gpurun_out/${1:-r06_cpu}; mkdir -p $OUT
for cores in 0-15 0-7 0-3 0-1; do
  DSV1_BENCH_NO_NUMA_PIN=1 taskset -c $cores python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --cpu-gops 4 > $OUT/c$cores.json 2> $OUT/c$cores.err
  python3 - $OUT/c$cores.json $cores <<'P'
import json,sys
p=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
sb=p["step_breakdown"]
print("cores", sys.argv[2], p["value"], p["ms_per_step"], "idle", sb["device_idle_ms_per_batch"], "fetch", sb["fetch_host_ms_per_call"], "host", {k.split(" (")[0]: v for k, v in sb["host_ms_per_batch"].items()})
P
done
