#!/bin/bash
mkdir -p gpurun_out
tools/r2_trace.sh w8 "COCONS_ENGINE=1"
f=$(find gpurun_out/r2_tr_w8 -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $f 400 > gpurun_out/r3_timeline_w8.txt 2>&1
tools/r3_run10.sh
