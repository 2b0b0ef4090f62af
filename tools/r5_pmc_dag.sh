#!/bin/bash
# Counter evidence for the dominant kernel (round 5): cocons::dag_kernel replayed alone (tools/dag_replay.py), one rocprofv3
# pass per counter group -- --pmc with --kernel-trace only, the program directly behind `--`.  Summary: tools/summarize_pmc_dag.py
R=$PWD
export TMPDIR=/tmp
cd /tmp
P="python3 $R/tools/dag_replay.py --reps 2 --warm 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5_dag_trace -o t -- $P > $R/gpurun_out/r5_dag_trace.log 2>&1
echo "trace rc=$?"
i=0
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $R/gpurun_out/r5_dag_pmc_$i -o p -- $P > $R/gpurun_out/r5_dag_pmc_$i.log 2>&1
  echo "pmc pass $i ($pmc) rc=$?"
done
