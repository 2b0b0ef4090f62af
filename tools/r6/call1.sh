#!/bin/bash
# round 6, call 1: do the fp64 vector and matrix pipes run side by side (cocons_corun_probe), and the round's baseline bench line
set -e
mkdir -p gpurun_out/r6
python tools/diag/corun.py > gpurun_out/r6/corun.txt 2>&1
cat gpurun_out/r6/corun.txt
python bench.py > gpurun_out/r6/bench_base.json 2> gpurun_out/r6/bench_base.err
tail -c 3000 gpurun_out/r6/bench_base.json
