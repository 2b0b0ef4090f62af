#!/bin/bash
mkdir -p gpurun_out
for rep in 1 2; do
  for w in new prev; do
    if [ $w = prev ]; then export COCONS_HIP_LIB=$PWD/tools/diag/prev/libcocons_hip.so; else unset COCONS_HIP_LIB; fi
    echo "== $w rep $rep"; COCONS_DEBUG_ABORT=1 timeout -k 10 200 python3 tools/diag/handle_churn.py 14 2>&1 | tail -18
  done
done 2>&1 | tee gpurun_out/r5_churn_ab.txt
