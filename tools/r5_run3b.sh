#!/bin/bash
mkdir -p gpurun_out
for a in "30 0 0" "30 0 1"; do
  COCONS_DEBUG_ABORT=1 timeout -k 5 60 python3 tools/diag/chain_probe.py $a 2>&1 | tail -25
  rc=$?; echo "== $a rc=$rc"
  [ $rc -eq 0 ] || exit 1
done 2>&1 | tee gpurun_out/r5_chain_probe.txt
