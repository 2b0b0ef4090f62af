#!/bin/bash
# Round-4 evidence: default bench line, 2-rank rehearsal of the sharded path, rocprofv3 kernel trace of the bench command
# (engine + DAG schedule on), and the per-task trace of the persistent launch (tools/dag_trace.py).
R=$PWD
mkdir -p $R/gpurun_out
python3 bench.py > $R/gpurun_out/r4_bench.json 2> $R/gpurun_out/r4_bench.err; echo "bench rc=$?"
COCONS_BENCH_REHEARSAL=1 timeout -k 10 500 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r4_bench_g2_rehearsal.json 2> $R/gpurun_out/r4_bench_g2.err; echo "bench --gpus 2 (rehearsal) rc=$?"
python3 tools/dag_trace.py --n 10000 --every 1 > $R/gpurun_out/r4_dag_trace_n10000.txt 2>&1; echo "dag trace rc=$?"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_prof_trace -o t -- python3 $R/bench.py --no-cpu-baseline --steps 15 --warmup 2 --inflight 0 > $R/gpurun_out/r4_prof_trace.log 2>&1
echo "trace rc=$?"
