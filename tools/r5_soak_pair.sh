#!/bin/bash
# hand-off time-outs of the pair engine: soak at n = 10^4 (default schedule), a loop at n = 4096, the chain layout, and a busy-host variant
N=${1:-3000}
mkdir -p gpurun_out
{
  echo "== n=10000 default"; COCONS_DEBUG_ABORT=1 timeout -k 10 300 python3 tools/soak.py $N 2>&1 | tail -4
  echo "== n=4096 default"; COCONS_DEBUG_ABORT=1 timeout -k 10 200 python3 tools/ab_modes.py --n 4096 --rounds 5 --evals 1000 "m0:" 2>&1 | tail -2
  echo "== n=10000 every step under the persistent launch"; COCONS_DEBUG_ABORT=1 COCONS_DAG_MIN_TILES=0 timeout -k 10 200 python3 tools/soak.py 1000 2>&1 | tail -4
  echo "== n=10000 chain layout"; COCONS_DEBUG_ABORT=1 COCONS_DAG_CHAIN=1 COCONS_DAG_MIN_TILES=0 timeout -k 10 200 python3 tools/soak.py 1000 2>&1 | tail -4
  echo "== n=2115 (a last block of one tile)"; COCONS_DEBUG_ABORT=1 timeout -k 10 200 python3 tools/ab_modes.py --n 2116 --rounds 5 --evals 1000 "m0:" 2>&1 | tail -2
} 2>&1 | tee gpurun_out/r5_soak_pair.txt
