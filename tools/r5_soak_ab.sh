#!/bin/bash
# hand-off time-outs: this tree's build against the round-4 build (tools/diag/prev) on ONE box, alternating, soak runs of
# N evaluations at n = 10^4 and a loop at n = 4096
N=${1:-1500}
mkdir -p gpurun_out
for rep in 1 2; do
  for w in new prev; do
    if [ $w = prev ]; then export COCONS_HIP_LIB=$PWD/tools/diag/prev/libcocons_hip.so; else unset COCONS_HIP_LIB; fi
    echo "== $w rep $rep n=10000"; COCONS_DEBUG_ABORT=1 timeout -k 10 200 python3 tools/soak.py $N 2>&1 | tail -4
    echo "== $w rep $rep n=4096"; COCONS_DEBUG_ABORT=1 timeout -k 10 100 python3 tools/ab_modes.py --n 4096 --rounds 3 --evals 400 "m0:" 2>&1 | tail -2
  done
done 2>&1 | tee gpurun_out/r5_soak_ab.txt
