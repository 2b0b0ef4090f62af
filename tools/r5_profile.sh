#!/bin/bash
# Round-5 evidence (one GPU call): default bench line, rocprofv3 kernel trace of the bench command (engine + DAG schedule on),
# per-task trace of the persistent launch (one-list layout) and of the chain layout, the replayed launch under the counters,
# kernel trace of a batch at n = 4096 (hardware queues), kernel stats of the sharded schedule on one rank (RCCL, second
# communicator).  Summaries: tools/r5_profile_collect.py
R=$PWD
mkdir -p $R/gpurun_out
python3 bench.py > $R/gpurun_out/r5_bench_final.json 2> $R/gpurun_out/r5_bench_final.err; echo "bench rc=$?"
python3 tools/dag_trace.py --n 10000 --every 1 > $R/gpurun_out/r5_dag_trace_n10000.txt 2>&1; echo "dag trace rc=$?"
python3 tools/chain_trace.py --n 10000 --min-tiles 0 --steps 2,14,22,30,36 > $R/gpurun_out/r5_chain_trace_n10000.txt 2>&1; echo "chain trace rc=$?"
python3 tools/batch_probe.py > $R/gpurun_out/r5_batch_probe_final.txt 2>&1; echo "batch probe rc=$?"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5_prof_trace -o t -- python3 $R/bench.py --no-cpu-baseline --no-configs --steps 15 --warmup 2 --inflight 0 > $R/gpurun_out/r5_prof_trace.log 2>&1
echo "bench trace rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5_shard_trace -o t -- python3 $R/tools/diag/rccl_one_rank.py 100 > $R/gpurun_out/r5_shard_trace.log 2>&1
echo "shard trace rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r5_bt_final -o t -- python3 $R/tools/diag/batch_trace.py 64 > $R/gpurun_out/r5_bt_final.log 2>&1
echo "batch trace rc=$?"; python3 $R/tools/diag/batch_trace_read.py $R/gpurun_out/r5_bt_final > $R/gpurun_out/r5_batch_trace_n4096.txt 2>&1; cat $R/gpurun_out/r5_batch_trace_n4096.txt
cd $R
bash tools/r5_pmc_dag.sh
