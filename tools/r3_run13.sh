#!/bin/bash
mkdir -p gpurun_out
COCONS_UPD_WAVES=8 COCONS_UPD_W8_MAX_TILES=3500 timeout -k 10 300 python -m pytest tests/test_gpu_engine_sizes.py tests/test_gpu_configs.py -m gpu -x -q -k "not shared and not worker and not native" > gpurun_out/r3_tests_w8.log 2>&1
echo "tests w8 rc=$?"; tail -2 gpurun_out/r3_tests_w8.log
python tools/ab_modes.py --rounds 9 "w4:upd_waves=4" "t3500:upd_waves=8,w8_max_tiles=3500,w8_inpanel=0" "t3500ip:upd_waves=8,w8_max_tiles=3500,w8_inpanel=1" 2>&1
python tools/ab_modes.py --n 4096 --rounds 9 --evals 20 "w4:upd_waves=4" "t3500:upd_waves=8,w8_max_tiles=3500,w8_inpanel=0" "t3500ip:upd_waves=8,w8_max_tiles=3500,w8_inpanel=1" 2>&1
