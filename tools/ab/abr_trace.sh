#!/bin/bash
# kernel trace of the drop-in ABR path (one picture coded at a time): kernel time against wall time per frame, launches per frame
REPO=$PWD; OUT=$REPO/gpurun_out/abr; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ak
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/ak -- python3 $REPO/tools/dropin_abr_fps.py 192 > $OUT/abr.txt 2>/dev/null
t=$(ls /tmp/ak/*/*kernel_trace.csv | head -1); python3 $REPO/tools/trace_summary.py "$t" > $OUT/abr_kernel_trace_summary.txt
cat $OUT/abr.txt; head -40 $OUT/abr_kernel_trace_summary.txt
python3 - "$t" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# take a window in the middle: 2000 kernels
w=rows[len(rows)//2:len(rows)//2+2000]
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in w)
span=int(w[-1]['End_Timestamp'])-int(w[0]['Start_Timestamp'])
print('window: %d kernels, span %.2f ms, kernel time %.2f ms (%.0f %%), mean kernel %.1f us, mean gap %.1f us' % (len(w), span/1e6, busy/1e6, 100*busy/span, busy/len(w)/1e3, (span-busy)/len(w)/1e3))
P
