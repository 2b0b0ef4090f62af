#!/bin/bash
# Parity suites under every documented non-default schedule switch: the alternatives that are kept must stay green.
# The whole matrix takes longer than one gpurun call may last: PART=1 / PART=2 run its two halves (results are appended to
# gpurun_out/switch_matrix.txt), PART=retry the configurations named in $RETRY (separated by ';').
mkdir -p gpurun_out
PART=${PART:-1}
part1=("COCONS_DAG=0" "COCONS_DAG_XCD=0" "COCONS_DAG_XCD=0 COCONS_DAG_ORDER=0" "COCONS_DAG_MIN_TILES=0" "COCONS_DAG_SPLIT=0" "COCONS_DAG_XCC_QUOTA=0" "COCONS_DAG_LEAD=3600 COCONS_DAG_LEAD2=0 COCONS_DAG_LEAD3=0"
       "COCONS_BATCH_ENGINE=0 COCONS_BATCH_SLOTS=3" "COCONS_ENGINE=0" "COCONS_UPD_WAVES=4" "COCONS_UPD_W8_MAX_TILES=0" "COCONS_UPD_DYNAMIC=0" "COCONS_FRONT_PAD=0")
part2=("COCONS_RHS_SLOTS=0" "COCONS_TAPER_PACKED=0" "COCONS_SPATIAL_SORT=0" "COCONS_PAIR_BLOCKED=0" "COCONS_TAPER_RCM=0" "COCONS_TAPER_BAND=0"
       "COCONS_BATCH_SLOTS=1" "COCONS_BATCH_SLOTS=4"
       "COCONS_ENGINE_PAIR=0" "COCONS_PANEL_FUSED=0" "COCONS_ENGINE_PAIR=0 COCONS_PANEL_FUSED=0 COCONS_DAG_MIN_TILES=0"
       "COCONS_POTRF_FOLLOW=0" "COCONS_POTRF_FOLLOW=0 COCONS_ENGINE=0" "COCONS_PANEL_FOLLOW=0"
       "COCONS_PANEL_DIAG=0" "COCONS_PANEL_SPLIT=0" "COCONS_PANEL_SPLIT=1"
       "COCONS_ENGINE_BLOCK0=0")
shard=()
case "$PART" in
  1) cfgs=("${part1[@]}"); : > gpurun_out/switch_matrix.txt ;;
  2) cfgs=("${part2[@]}"); shard=("COCONS_SHARD_GROUP=1" "COCONS_SHARD_GROUP=2" "COCONS_SHARD_COMM2=1") ;;
  retry) IFS=';' read -r -a cfgs <<< "$RETRY"; shard=(); IFS=';' read -r -a shard <<< "${RETRY_SHARD:-}" ;;
  *) echo "PART=1|2|retry"; exit 2 ;;
esac
for cfg in "${cfgs[@]}"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  env $cfg timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine_sizes.py tests/test_gpu_configs.py tests/test_gpu_dag.py -x -q -m gpu \
      -k "not sharded and not worker and not split_two" > gpurun_out/switch_$tag.log 2>&1
  rc=$?
  echo "$cfg rc=$rc $(tail -1 gpurun_out/switch_$tag.log)" | tee -a gpurun_out/switch_matrix.txt
  if [ $rc -ge 124 ]; then echo "timed out: stopping"; exit 1; fi
done
# the deal of the sharded evaluation (read once per process: the shared-GPU cases pass their own group; here the default changes)
for cfg in "${shard[@]}"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  env $cfg timeout -k 10 400 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "sharded" > gpurun_out/switch_$tag.log 2>&1
  rc=$?
  echo "$cfg (sharded cases) rc=$rc $(tail -1 gpurun_out/switch_$tag.log)" | tee -a gpurun_out/switch_matrix.txt
  if [ $rc -ge 124 ]; then echo "timed out: stopping"; exit 1; fi
done
