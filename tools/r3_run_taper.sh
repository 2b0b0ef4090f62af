#!/bin/bash
mkdir -p gpurun_out
for g in 100 200 316; do
  timeout -k 10 500 python tools/taper_timing.py $g $(python3 -c "print(6.0/$g)") nocpu > gpurun_out/r3_taper_$g.log 2>&1
  echo "g=$g rc=$?"; tail -4 gpurun_out/r3_taper_$g.log
done
