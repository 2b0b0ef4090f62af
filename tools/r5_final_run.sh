#!/bin/bash
# final validation of the round on one box: the switch configurations that touch the kernels changed last, soak, bench line
mkdir -p gpurun_out
PART=retry RETRY="COCONS_ENGINE_BLOCK0=0;COCONS_DAG=0;COCONS_DAG_MIN_TILES=0;COCONS_ENGINE_PAIR=0;COCONS_PANEL_FUSED=0;COCONS_DAG_CHAIN=1 COCONS_DAG_MIN_TILES=0;COCONS_FRONT_PAD=0;COCONS_RHS_SLOTS=0;COCONS_PANEL_DIAG=0" RETRY_SHARD="COCONS_SHARD_GROUP=1" bash tools/gpu_switch_matrix.sh
bash tools/r5_soak_pair.sh 2000 > /dev/null 2>&1; grep -E "^==|done:|engine retries|median" gpurun_out/r5_soak_pair.txt
python bench.py > gpurun_out/r5_bench_final.json 2> gpurun_out/r5_bench_final.err; echo "bench rc=$?"
