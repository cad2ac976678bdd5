#!/bin/bash
# round 6: the new GPU tests + the driver's exact bench command; output under gpurun_out/$1
TAG=${1:-r06a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cat /proc/loadavg > $OUT/loadavg.txt
timeout 1500 python3 -m pytest tests/test_gpu_stream.py tests/test_bench_launcher.py tests/test_gpu_bench_contract.py tests/test_gpu_pipeline_abi.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python3 - $OUT/bench.json <<'P'
import json,sys
p=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(p["value"], p["ms_per_step"], p["link"], p["value_host_pinned"])
print(json.dumps(p["step_breakdown"], indent=1))
print(json.dumps(p["box"], indent=1))
print(p["value_recon_all"])
print({k: (v.get("Mpix_s"), v.get("ms_per_step"), v.get("bit_exact_vs_cpu")) if isinstance(v, dict) else v for k, v in p["shapes"].items()})
P
