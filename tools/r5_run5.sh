#!/bin/bash
mkdir -p gpurun_out
for w in new prev; do
  if [ $w = prev ]; then export COCONS_HIP_LIB=$PWD/tools/diag/prev/libcocons_hip.so; else unset COCONS_HIP_LIB; fi
  echo "== $w"
  timeout -k 5 120 python3 tools/chain_trace.py --n 10000 --min-tiles 0 --steps 30 2>&1 | tail -8
  COCONS_DEBUG_ABORT=1 timeout -k 10 300 python3 tools/ab_modes.py --n 10000 --rounds 5 --evals 20 "old:dag_chain=0" "chain_all:dag_min_tiles=0" "chain_800:dag_min_tiles=800" 2>&1 | tail -4
  COCONS_DEBUG_ABORT=1 timeout -k 10 200 python3 tools/ab_modes.py --n 4096 --rounds 5 --evals 60 "classic:" "dag_all:dag_min_tiles=0" 2>&1 | tail -3
done 2>&1 | tee gpurun_out/r5_run5.txt
