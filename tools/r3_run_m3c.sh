#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python tools/ab_modes.py --rounds 7 --evals 10 "m0:" "m3:panel_mode=3" "m3_4500:panel_mode=3,strip_min=4500" > gpurun_out/r3_m3_ab.log 2>&1
echo "ab rc=$?"; tail -5 gpurun_out/r3_m3_ab.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in 0 3; do
  export COCONS_PANEL_MODE=$m
  timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r3_tr_m$m -o tr --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 2 --inflight 0 > $R/gpurun_out/r3_tr_m$m.log 2>&1 || exit 1
  f=$(ls $R/gpurun_out/r3_tr_m$m/*kernel_trace.csv $R/gpurun_out/r3_tr_m$m/*/*kernel_trace.csv 2>/dev/null | head -1)
  python3 $R/tools/trace_timeline.py $f 200 > $R/gpurun_out/r3_timeline_sb_m$m.txt 2>&1
  rm -rf $R/gpurun_out/r3_tr_m$m
done
echo done
