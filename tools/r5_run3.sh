#!/bin/bash
# chain helpers: correctness first, then variants in alternation
mkdir -p gpurun_out
COCONS_DEBUG_ABORT=1 timeout -k 10 300 python -m pytest tests/test_gpu_dag.py tests/test_gpu_parity.py -m gpu -x -q -k "dag or timeout or neg2loglik_vs_cpu or fuzz" > gpurun_out/r5_dagtest.log 2>&1; rc=$?; tail -15 gpurun_out/r5_dagtest.log; echo "dag tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
COCONS_DEBUG_ABORT=1 timeout -k 10 300 python3 tools/ab_modes.py --n 10000 --rounds 5 --evals 20 "old:dag_chain=0" "chain:" "chain_all:dag_min_tiles=0" "chain_1500:dag_min_tiles=1500" "chain_all_h16:dag_min_tiles=0,dag_helpers=16" 2>&1 | tail -8 | tee gpurun_out/r5_ab_chain.txt
COCONS_DEBUG_ABORT=1 timeout -k 10 200 python3 tools/ab_modes.py --n 4096 --rounds 5 --evals 60 "classic:" "dag_all:dag_min_tiles=0" 2>&1 | tail -4 | tee -a gpurun_out/r5_ab_chain.txt
