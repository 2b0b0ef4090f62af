#!/bin/bash
python tools/ab_modes.py --rounds 9 "m0f1:panel_mode=0,engine_fused=1" "m0f0:panel_mode=0,engine_fused=0" "m2f1:panel_mode=2,engine_fused=1" "m2f0:panel_mode=2,engine_fused=0" "m2l2:panel_mode=2,strip_lead=2000" "m1:panel_mode=1" 2>&1
python tools/ab_modes.py --n 4096 --rounds 9 --evals 20 "m0f1:panel_mode=0,engine_fused=1" "m0f0:panel_mode=0,engine_fused=0" "m2f1:panel_mode=2,strip_min=500" 2>&1
