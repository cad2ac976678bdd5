#!/bin/bash
# round 6, verdict item 1: the driver's exact command on this box, several times, with the host-phase breakdown, and the same with the
# NUMA pin off / forced to each node.  Output: gpurun_out/$TAG/
TAG=${1:-r06_repro}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
{
  echo "== loadavg"; cat /proc/loadavg
  echo "== nproc $(nproc)"; lscpu | grep -E "Model name|Socket|NUMA|Thread|MHz" 
  echo "== affinity"; python3 -c "import os; print(len(os.sched_getaffinity(0)))"
  echo "== gpus"; for d in /sys/bus/pci/devices/*; do v=$(cat $d/vendor 2>/dev/null); c=$(cat $d/class 2>/dev/null); if [ "$v" = "0x1002" ]; then echo "$d class $c numa $(cat $d/numa_node) speed $(cat $d/current_link_speed 2>/dev/null) width $(cat $d/current_link_width 2>/dev/null)"; fi; done
  echo "== visible"; env | grep -E "VISIBLE|HSA_|GPU_|ROCR" 
  echo "== top cpu"; ps -eo pcpu,pid,comm --sort=-pcpu | head -8
  echo "== meminfo"; grep -E "MemTotal|MemFree|MemAvailable|HugePages_Total" /proc/meminfo
  echo "== cgroup cpu"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null
} > "$OUT/box.txt" 2>&1
run() { # name, env...
  n=$1; shift
  env "$@" DSV1_HOST_PROF=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > "$OUT/$n.json" 2> "$OUT/$n.err"
  python3 - "$OUT/$n.json" "$n" <<'P'
import json,sys
try:
    p=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], p["value"], p["ms_per_step"], p["config"].get("numa_node_of_gpu_rank0"), p["config"].get("host_cores_rank0"), p.get("timed_region"), (p.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[2], "failed", e)
P
}
run a1
run a2
run nopin DSV1_BENCH_NO_NUMA_PIN=1
run node0 DSV1_BENCH_NUMA_NODE=0
run node1 DSV1_BENCH_NUMA_NODE=1
cat /proc/loadavg
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/full.json" 2> "$OUT/full.err"
tail -c 3000 "$OUT/full.json"
grep "dsv1 host" "$OUT/a1.err" "$OUT/node0.err" "$OUT/node1.err"
