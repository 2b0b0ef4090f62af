#!/bin/bash
# first GPU run of the engine's pair mode: correctness on a few sizes, then A/B against the one-workgroup engine
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_engine_sizes.py tests/test_gpu_dag.py -x -q -m gpu > gpurun_out/pair_tests.log 2>&1
rc=$?; tail -5 gpurun_out/pair_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python tools/ab_modes.py --n 10000 --rounds 5 --evals 10 "pair:engine_pair=1" "single:engine_pair=0" > gpurun_out/pair_ab_n10000.txt 2>&1 || exit 1
tail -6 gpurun_out/pair_ab_n10000.txt
timeout -k 10 200 python tools/ab_modes.py --n 4096 --rounds 5 --evals 30 "pair:engine_pair=1" "single:engine_pair=0" > gpurun_out/pair_ab_n4096.txt 2>&1 || exit 1
tail -6 gpurun_out/pair_ab_n4096.txt
