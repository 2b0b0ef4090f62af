#!/bin/bash
# round 5, second GPU call: the tests that changed, the replayed launch and its counter passes, batch throughput against the
# number of hardware queues
mkdir -p gpurun_out
COCONS_DEBUG_ABORT=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_glue_exec.py -m gpu -q -k "degenerate or glue or cached or cov_entries" > gpurun_out/r5_gputest2.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/r5_gputest2.log)"
timeout -k 10 120 python3 tools/dag_replay.py --reps 3 > gpurun_out/r5_replay.json 2> gpurun_out/r5_replay.err; rc2=$?; cat gpurun_out/r5_replay.json; tail -3 gpurun_out/r5_replay.err; echo "replay rc=$rc2"
for q in "" 8 16; do
  for cfg in "" "COCONS_BATCH_SLOTS=3" "COCONS_BATCH_ENGINE=0 COCONS_BATCH_SLOTS=3"; do
    env ${q:+GPU_MAX_HW_QUEUES=$q} $cfg timeout -k 10 120 python3 tools/batch_probe.py 2>&1 | grep "n=" 
  done
done | tee gpurun_out/r5_batch_probe.txt
[ $rc2 -eq 0 ] && bash tools/r5_pmc_dag.sh
