#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python tools/ab_modes.py --rounds 5 --evals 10 "m0:" "m3:panel_mode=3" "m3_5000:panel_mode=3,strip_min=5000" "m3_2500:panel_mode=3,strip_min=2500" > gpurun_out/r3_m3_ab.log 2>&1
echo "ab rc=$?"; tail -6 gpurun_out/r3_m3_ab.log
cd /tmp && export TMPDIR=/tmp
export COCONS_PANEL_MODE=3
timeout -k 10 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r3_tr_m3 -o tr --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 2 --inflight 0 > $GRAFT_REPO_ROOT/gpurun_out/r3_tr_m3.log 2>&1
echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/r3_tr_m3/*kernel_trace.csv gpurun_out/r3_tr_m3/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 tools/trace_timeline.py $f 200 > gpurun_out/r3_timeline_m3.txt 2>&1
rm -rf gpurun_out/r3_tr_m3
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine_sizes.py -x -q -m gpu > gpurun_out/r3_m3_tests.log 2>&1
echo "tests(mode3) rc=$?"; tail -3 gpurun_out/r3_m3_tests.log
