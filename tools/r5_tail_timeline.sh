#!/bin/bash
# kernel timeline of one evaluation at n = 10^4 (default schedule): what the main stream does in the tail of the factorisation
R=$PWD
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r5_tl -o t -- python3 $R/bench.py --no-cpu-baseline --no-configs --steps 6 --warmup 2 --inflight 0 > $R/gpurun_out/r5_tl.log 2>&1
echo "trace rc=$?"
cd $R
f=$(find gpurun_out/r5_tl -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $f 400 > gpurun_out/r5_tail_timeline.txt 2>&1
tail -150 gpurun_out/r5_tail_timeline.txt
