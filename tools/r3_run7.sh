#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_engine_sizes.py tests/test_gpu_configs.py -m gpu -x -q -k "not shared and not worker and not native" > gpurun_out/r3_tests_a.log 2>&1
echo "tests-a rc=$?"; tail -3 gpurun_out/r3_tests_a.log
python tools/ab_modes.py --rounds 7 "f0:engine_fused=0" "f1:engine_fused=1" > gpurun_out/r3_abm5.log 2>&1; cat gpurun_out/r3_abm5.log
python tools/ab_modes.py --n 4096 --rounds 7 --evals 20 "f0:engine_fused=0" "f1:engine_fused=1" > gpurun_out/r3_abm5b.log 2>&1; cat gpurun_out/r3_abm5b.log
