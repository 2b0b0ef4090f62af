#!/bin/bash
mkdir -p gpurun_out
COCONS_DEBUG_ABORT=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_engine_sizes.py -m gpu -x -q -k "batch or Hessian or replica or c4 or taper_objective" > gpurun_out/r5_batchtest.log 2>&1; rc=$?; tail -5 gpurun_out/r5_batchtest.log; echo "batch tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
for cfg in "" "COCONS_BATCH_ENGINE=0 COCONS_BATCH_SLOTS=3" "COCONS_BATCH_ENGINE=0 COCONS_BATCH_SLOTS=4" "COCONS_BATCH_ENGINE=0 COCONS_BATCH_SLOTS=6" "COCONS_BATCH_ENGINE=1 COCONS_BATCH_SLOTS=2"; do
  env $cfg timeout -k 10 120 python3 tools/batch_probe.py 2>&1 | grep "n="
done | tee gpurun_out/r5_batch_probe3.txt
