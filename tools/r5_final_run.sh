#!/bin/bash
# final validation of the round on one box: full GPU suite, the switch configurations that touch the kernels changed last, soak
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/full_gpu.log 2>&1; echo "full suite rc=$? $(tail -1 gpurun_out/full_gpu.log)"
PART=retry RETRY="COCONS_PANEL_SPLIT=0;COCONS_PANEL_DIAG=0;COCONS_PANEL_FOLLOW=0;COCONS_ENGINE_PAIR=0;COCONS_UPD_DYNAMIC=0;COCONS_DAG_MIN_TILES=0;COCONS_RHS_SLOTS=0" RETRY_SHARD="" bash tools/gpu_switch_matrix.sh
bash tools/r5_soak_pair.sh 2000 > /dev/null 2>&1; grep -E "^==|done:|engine retries|median" gpurun_out/r5_soak_pair.txt
