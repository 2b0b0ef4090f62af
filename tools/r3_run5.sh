#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > gpurun_out/r3_tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|worker |the same" gpurun_out/r3_tests.log | tail -8
