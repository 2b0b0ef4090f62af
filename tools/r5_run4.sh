#!/bin/bash
mkdir -p gpurun_out
COCONS_DEBUG_ABORT=1 timeout -k 10 300 python -m pytest tests/test_gpu_dag.py tests/test_gpu_parity.py -m gpu -x -q -k "dag or timeout or neg2loglik_vs_cpu or fuzz" > gpurun_out/r5_dagtest.log 2>&1; rc=$?; tail -5 gpurun_out/r5_dagtest.log; echo "dag tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 5 120 python3 tools/chain_trace.py --n 10000 --min-tiles 0 --steps 14,30,36 > gpurun_out/r5_chain_trace2.txt 2>&1; tail -22 gpurun_out/r5_chain_trace2.txt
COCONS_DEBUG_ABORT=1 timeout -k 10 300 python3 tools/ab_modes.py --n 10000 --rounds 5 --evals 20 "old:dag_chain=0" "chain:" "chain_all:dag_min_tiles=0" "chain_1500:dag_min_tiles=1500" "chain_800:dag_min_tiles=800" 2>&1 | tail -8 | tee gpurun_out/r5_ab_chain.txt
COCONS_DEBUG_ABORT=1 timeout -k 10 200 python3 tools/ab_modes.py --n 4096 --rounds 5 --evals 60 "classic:" "dag_all:dag_min_tiles=0" 2>&1 | tail -4 | tee -a gpurun_out/r5_ab_chain.txt
