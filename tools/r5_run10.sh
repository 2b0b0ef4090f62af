#!/bin/bash
mkdir -p gpurun_out
COCONS_DEBUG_ABORT=1 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine_sizes.py -m gpu -x -q > gpurun_out/r5_fusetest.log 2>&1; rc=$?; tail -4 gpurun_out/r5_fusetest.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
for cfg in "COCONS_FUSE_PANEL=0" "COCONS_FUSE_PANEL=1"; do
  echo "== $cfg"
  env $cfg timeout -k 10 120 python3 tools/batch_probe.py 2>&1 | grep "n="
  env $cfg timeout -k 10 200 python3 tools/taper_timing.py 2>&1 | tail -4
done 2>&1 | tee gpurun_out/r5_fuse_ab.txt
