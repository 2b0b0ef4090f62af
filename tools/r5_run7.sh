#!/bin/bash
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
for cfg in "COCONS_BATCH_SLOTS=2" "COCONS_BATCH_SLOTS=4" "COCONS_BATCH_ENGINE=0 COCONS_BATCH_SLOTS=4"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  export $cfg
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r5_bt_$tag -o t -- python3 $R/tools/diag/batch_trace.py 64 > $R/gpurun_out/r5_bt_$tag.log 2>&1
  echo "== $cfg rc=$?"; python3 $R/tools/diag/batch_trace_read.py $R/gpurun_out/r5_bt_$tag
  unset COCONS_BATCH_SLOTS COCONS_BATCH_ENGINE
done 2>&1 | tee $R/gpurun_out/r5_batch_trace.txt
