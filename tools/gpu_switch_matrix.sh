#!/bin/bash
# Parity suites under every documented non-default schedule switch: the alternatives that are kept must stay green.
mkdir -p gpurun_out
: > gpurun_out/switch_matrix.txt
for cfg in "COCONS_DAG=0" "COCONS_DAG_MIN_TILES=0" "COCONS_DAG_SPLIT=0" "COCONS_DAG_XCC_QUOTA=0" "COCONS_DAG_LEAD=3600 COCONS_DAG_LEAD2=0 COCONS_DAG_LEAD3=0" "COCONS_BATCH_ENGINE=0 COCONS_BATCH_SLOTS=3" "COCONS_ENGINE=0" "COCONS_UPD_WAVES=4" "COCONS_UPD_W8_MAX_TILES=0" "COCONS_UPD_DYNAMIC=0" \
           "COCONS_FRONT_PAD=0" "COCONS_RHS_SLOTS=0" "COCONS_TAPER_PACKED=0" "COCONS_SPATIAL_SORT=0" "COCONS_PAIR_BLOCKED=0" \
           "COCONS_TAPER_RCM=0" "COCONS_TAPER_BAND=0" "COCONS_BATCH_SLOTS=1" "COCONS_BATCH_SLOTS=4" "COCONS_DAG_CHAIN=1" "COCONS_DAG_CHAIN=1 COCONS_DAG_MIN_TILES=0"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  env $cfg timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine_sizes.py tests/test_gpu_configs.py tests/test_gpu_dag.py -x -q -m gpu \
      -k "not sharded and not worker and not split_two" > gpurun_out/switch_$tag.log 2>&1
  rc=$?
  echo "$cfg rc=$rc $(tail -1 gpurun_out/switch_$tag.log)" | tee -a gpurun_out/switch_matrix.txt
  if [ $rc -ge 124 ]; then echo "timed out: stopping"; exit 1; fi
done
# the deal of the sharded evaluation (read once per process: the shared-GPU cases pass their own group; here the default changes)
for cfg in "COCONS_SHARD_GROUP=1" "COCONS_SHARD_GROUP=2" "COCONS_SHARD_COMM2=0"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  env $cfg timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "sharded" > gpurun_out/switch_$tag.log 2>&1
  rc=$?
  echo "$cfg (sharded cases) rc=$rc $(tail -1 gpurun_out/switch_$tag.log)" | tee -a gpurun_out/switch_matrix.txt
  if [ $rc -ge 124 ]; then echo "timed out: stopping"; exit 1; fi
done
