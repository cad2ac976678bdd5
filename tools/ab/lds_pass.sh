#!/bin/bash
# LDS bank conflicts per kernel of the headline workload (64 GOPs): cycles lost to conflicts against cycles with an LDS instruction active
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
PB="python3 $REPO/bench.py --cpu-gops 0 --steps 1 --warmup 1 --gops 64 --prof-kernel none --no-extras"
rm -rf /tmp/pl
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d /tmp/pl -- $PB > /dev/null 2>&1
python3 $REPO/tools/pmc_summary.py "/tmp/pl/**/*counter_collection.csv"
