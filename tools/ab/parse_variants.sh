#!/bin/bash
# timing experiment: k_hz_parse with parts disabled (PV=1: no code decode, 2: no run->position pass, 3: neither)
R=$PWD
cd /tmp && export TMPDIR=/tmp
for pv in 0 1 2 3; do
  touch $R/digital-subband-video-1_amd/csrc/k_hzcc.hip
  make -C $R/digital-subband-video-1_amd/csrc -j8 EXTRA=-DPV=$pv > /dev/null 2>&1
  rm -rf /tmp/kd$pv
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kd$pv -- python3 $R/tools/decode_time.py > /dev/null 2>&1
  t=$(ls /tmp/kd$pv/*/*kernel_trace.csv | head -1)
  echo "PV=$pv"; python3 $R/tools/trace_summary.py "$t" k_hz_parse | tail -1
done
