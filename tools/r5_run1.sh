#!/bin/bash
# round 5, first GPU call: the whole -m gpu suite, the replayed launch once, the default bench line
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r5_gputest.log 2>&1; rc=$?; tail -5 gpurun_out/r5_gputest.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python3 tools/dag_replay.py --reps 3 > gpurun_out/r5_replay.json 2> gpurun_out/r5_replay.err; rc=$?; cat gpurun_out/r5_replay.json; tail -3 gpurun_out/r5_replay.err; echo "replay rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err; rc=$?; tail -3 gpurun_out/r5_bench.err; echo "bench rc=$rc"; head -c 6000 gpurun_out/r5_bench.json
exit $rc
