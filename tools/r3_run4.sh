#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r3_tests.log
tools/r3_ab.sh m2 "COCONS_PANEL_MODE=2" m0 "COCONS_PANEL_MODE=0" m2l2 "COCONS_STRIP_LEAD=2000" m2l6 "COCONS_STRIP_LEAD=6000" m2min "COCONS_STRIP_MIN=1500" m2b "COCONS_PANEL_MODE=2" m0b "COCONS_PANEL_MODE=0"
tools/r2_trace.sh m2 "COCONS_PANEL_MODE=2"
f=$(find gpurun_out/r2_tr_m2 -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $f 400 > gpurun_out/r3_timeline_m2.txt 2>&1
echo "timeline rc=$?"
