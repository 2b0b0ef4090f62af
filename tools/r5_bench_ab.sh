#!/bin/bash
# the default bench line of this tree's build against the round-4 build (tools/diag/prev), alternating processes on one box
mkdir -p gpurun_out
for rep in 1 2 3; do
  for w in prev new; do
    if [ $w = prev ]; then export COCONS_HIP_LIB=$PWD/tools/diag/prev/libcocons_hip.so; else unset COCONS_HIP_LIB; fi
    timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-configs --inflight 0 --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$w', d['value'], d['ms_per_step'], d['stages_ms']['eval_ms'], d['cholesky_frac'])"
  done
done 2>&1 | tee gpurun_out/r5_bench_ab.txt
