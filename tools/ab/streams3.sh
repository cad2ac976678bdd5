#!/bin/bash
# three coding streams with the fetch on the analysis stream (four hardware queues per process: does the third coding
# stream get one of its own then?)
one() { python bench.py --cpu-gops 0 --steps 6 --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['config'].get('streams_on_own_hw_queue'))"; }
one base; one base
export DSV1_FETCH_ON_ANALYSIS=1; one fetch_on_analysis; one fetch_on_analysis
export DSV1_CODE_STREAMS=3; one fetch_on_analysis+3streams; one fetch_on_analysis+3streams
