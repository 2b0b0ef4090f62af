#!/bin/bash
# kernel timeline of one evaluation (default schedule): what the main stream does in the tail of the factorisation
# usage: tools/r5_tail_timeline.sh [n = 10000] [tag]   (environment switches apply)
R=$PWD
N=${1:-10000}; TAG=${2:-tl}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r5_$TAG -o t -- python3 $R/bench.py --n $N --no-cpu-baseline --no-configs --steps 6 --warmup 2 --inflight 0 > $R/gpurun_out/r5_$TAG.log 2>&1
echo "trace rc=$?"
cd $R
f=$(find gpurun_out/r5_$TAG -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $f 400 > gpurun_out/r5_timeline_$TAG.txt 2>&1
