#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "not sharded and not worker" > gpurun_out/r3_asm_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r3_asm_tests.log
timeout -k 10 300 python tools/range_sweep.py > gpurun_out/r3_range_sweep.log 2>&1
echo "range rc=$?"; cat gpurun_out/r3_range_sweep.log
