#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "taper" > gpurun_out/r3_tests_t.log 2>&1
echo "taper tests rc=$?"; tail -3 gpurun_out/r3_tests_t.log
timeout -k 10 200 python tools/taper_timing.py 100 0.06 > gpurun_out/r3_taper_100.log 2>&1; tail -4 gpurun_out/r3_taper_100.log
COCONS_TAPER_PACKED=0 timeout -k 10 200 python tools/taper_timing.py 100 0.06 > gpurun_out/r3_taper_100_dense.log 2>&1; tail -3 gpurun_out/r3_taper_100_dense.log
timeout -k 10 500 python tools/taper_timing.py 316 0.019 > gpurun_out/r3_taper_316.log 2>&1; tail -4 gpurun_out/r3_taper_316.log
