#!/bin/bash
mkdir -p gpurun_out
COCONS_DEBUG_ABORT=1 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine_sizes.py tests/test_gpu_dag.py -m gpu -x -q > gpurun_out/r5_t12.log 2>&1; rc=$?; tail -3 gpurun_out/r5_t12.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
  for w in prev new; do
    if [ $w = prev ]; then export COCONS_HIP_LIB=$PWD/tools/diag/prev/libcocons_hip.so; else unset COCONS_HIP_LIB; fi
    echo "$w n=10000 $(timeout -k 10 120 python3 tools/ab_modes.py --n 10000 --rounds 3 --evals 30 'm0:' 2>&1 | tail -1 | cut -c1-120)"
    echo "$w n=4096  $(timeout -k 10 120 python3 tools/ab_modes.py --n 4096 --rounds 3 --evals 100 'm0:' 2>&1 | tail -1 | cut -c1-120)"
  done
done 2>&1 | tee gpurun_out/r5_ab12.txt
