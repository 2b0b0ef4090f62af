#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r3_tests.log
python tools/ab_modes.py --rounds 7 "s0:panel_mode=0,tail_split=0" "s1k:panel_mode=0,tail_split=1020" "s2k:panel_mode=0,tail_split=2040" "s4k:panel_mode=0,tail_split=4080" "s8k:panel_mode=0,tail_split=8160" "m2s2k:panel_mode=2,tail_split=2040" > gpurun_out/r3_abm4.log 2>&1; cat gpurun_out/r3_abm4.log
