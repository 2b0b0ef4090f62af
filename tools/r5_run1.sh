#!/bin/bash
# round 5, first GPU call: the whole -m gpu suite, the replayed launch once, the default bench line
mkdir -p gpurun_out
COCONS_DEBUG_ABORT=1 timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/r5_gputest.log 2>&1; rc=$?; tail -8 gpurun_out/r5_gputest.log; echo "pytest rc=$rc"
timeout -k 10 120 python3 tools/dag_replay.py --reps 3 > gpurun_out/r5_replay.json 2> gpurun_out/r5_replay.err; rc2=$?; cat gpurun_out/r5_replay.json; tail -3 gpurun_out/r5_replay.err; echo "replay rc=$rc2"
timeout -k 10 300 python3 bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err; rc3=$?; tail -3 gpurun_out/r5_bench.err; echo "bench rc=$rc3"; head -c 7000 gpurun_out/r5_bench.json
exit $rc
