#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r3_tests.log
tools/r3_ab.sh ov1 "COCONS_PANEL_OVERLAP=1" ov0 "COCONS_PANEL_OVERLAP=0" ov1b "COCONS_PANEL_OVERLAP=1" ov0b "COCONS_PANEL_OVERLAP=0" plain "COCONS_ENGINE=0"
tools/r2_trace.sh ov1 "COCONS_PANEL_OVERLAP=1"
f=$(find gpurun_out/r2_tr_ov1 -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $f 400 > gpurun_out/r3_timeline_ov1.txt 2>&1
echo "timeline rc=$?"
