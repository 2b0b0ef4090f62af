#!/bin/bash
# full GPU suite + the driver's bench command
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tests_full.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r3_tests_full.log
timeout -k 10 400 python bench.py > gpurun_out/r3_bench_full.log 2>&1
echo "bench rc=$?"; tail -1 gpurun_out/r3_bench_full.log | cut -c1-1500
