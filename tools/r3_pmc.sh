#!/bin/bash
# rocprofv3 passes behind profiles/r03_*: kernel trace of the default bench, then PMC passes (kernels are
# serialised under --pmc, so the resident engine is switched off there: its partner kernels could not run).
R=$PWD
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 --inflight 0"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_prof_trace -o t -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 --inflight 0 > $R/gpurun_out/r3_prof_trace.log 2>&1
echo "trace rc=$?"
export COCONS_ENGINE=0
i=0
for pmc in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $R/gpurun_out/r3_pmc_$i -o p -- $B > $R/gpurun_out/r3_pmc_$i.log 2>&1
  echo "pmc pass $i ($pmc) rc=$?"
done
