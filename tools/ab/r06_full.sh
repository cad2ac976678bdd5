#!/bin/bash
# round 6: the driver's exact bench command, then the whole GPU suite; output under gpurun_out/$1
TAG=${1:-r06d}; OUT=gpurun_out/$TAG; mkdir -p $OUT
cat /proc/loadavg > $OUT/loadavg.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python3 - $OUT/bench.json <<'P'
import json,sys
p=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(p["value"], p["ms_per_step"], p["link"], p["config"]["host_threads_rank0"], "idle", p["step_breakdown"]["device_idle_ms_per_batch"], "exact", p["bit_exact_vs_cpu"])
print({k: v.get("Mpix_s") for k, v in p["shapes"].items() if isinstance(v, dict)})
print(p["value_host_pinned"], p["value_recon_all"])
P
if [ "$2" != "nobench2" ]; then timeout 2400 python3 -m pytest tests -q -m gpu -x > $OUT/pytest.txt 2>&1; grep -E "passed|failed" $OUT/pytest.txt | tail -2; fi
