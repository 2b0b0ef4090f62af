#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 800 python tools/ab_modes.py --rounds 4 --evals 300 "m3:" "m0:panel_mode=0,strip_min=3600" > gpurun_out/r3_ab_long.log 2>&1
echo "rc=$?"; tail -4 gpurun_out/r3_ab_long.log
timeout -k 10 800 python tools/ab_modes.py --rounds 4 --evals 300 "m0:panel_mode=0,strip_min=3600" "m3:" > gpurun_out/r3_ab_long2.log 2>&1
echo "rc=$?"; tail -4 gpurun_out/r3_ab_long2.log
