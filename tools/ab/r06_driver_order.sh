#!/bin/bash
# the driver's round-end sequence on a fresh box: the GPU suite, smoke(), then the bench line
OUT=gpurun_out/${1:-r06g}; mkdir -p $OUT
timeout 2400 python3 -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -E "passed|failed" $OUT/pytest.txt | tail -1
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python3 - $OUT/bench.json <<'P'
import json,sys
p=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(p["value"], p["ms_per_step"], "idle", p["step_breakdown"]["device_idle_ms_per_batch"], "exact", p["bit_exact_vs_cpu"], "threads", p["config"]["host_threads_rank0"], p["box"]["loadavg"], p["box"]["cgroup_cpu_max"])
print({k: v.get("Mpix_s") for k, v in p["shapes"].items() if isinstance(v, dict)})
P
