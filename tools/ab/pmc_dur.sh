#!/bin/bash
# one counter pass WITH the kernel trace of the same run: are GRBM_GUI_ACTIVE / SQ_BUSY_CYCLES consistent with the dispatch durations?
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$OLDPWD}
OUT=$REPO/gpurun_out/clk
mkdir -p $OUT
PB="python3 $REPO/bench.py --cpu-gops 0 --steps 1 --warmup 1 --gops 160 --prof-kernel none --no-extras"
rm -rf /tmp/pa
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pa -- $PB > $OUT/pa.log 2>&1
ls -R /tmp/pa | head -20
f=$(ls /tmp/pa/*/*counter_collection.csv | head -1)
[ -n "$f" ] || { tail -5 $OUT/pa.log; exit 1; }
head -3 $f
grep "k_inv_p_tile<true>" $f | head -8 > $OUT/cc_invp.csv
grep "k_hme_level<true, 12" $f | head -24 > $OUT/cc_hme.csv
k=$(ls /tmp/pa/*/*kernel_trace.csv | head -1)
head -2 $k
grep "k_inv_p_tile<true>" $k | head -6 > $OUT/kt_invp.csv
grep "k_hme_level<true, 12" $k | head -6 > $OUT/kt_hme.csv
