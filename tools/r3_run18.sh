#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r3_tests.log
