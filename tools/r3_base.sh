#!/bin/bash
# round-3 baseline on a fresh box: GPU tests, the default bench, a kernel trace of the bench
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r3_tests.log
timeout -k 10 300 python bench.py > gpurun_out/r3_bench.log 2>&1
echo "bench rc=$?"; tail -1 gpurun_out/r3_bench.log
tools/r2_trace.sh base "COCONS_ENGINE=1"
f=$(find gpurun_out/r2_tr_base -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $f 400 > gpurun_out/r3_timeline_base.txt 2>&1
echo "timeline rc=$?"
