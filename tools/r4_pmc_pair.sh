#!/bin/bash
# rocprofv3 --pmc passes of the assembly kernel (pair_sym_kernel<0,false>) at correlation range 0.05 and 1.0, n = 10^4
# (kernels are serialised under --pmc: the resident engine is switched off).  Summary: tools/summarize_pmc_pair.py
R=$PWD
export TMPDIR=/tmp COCONS_ENGINE=0
cd /tmp
for rg in 0.05 1.0; do
  i=0
  for pmc in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
             "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $R/gpurun_out/r4_pair_${rg}_$i -o p -- python3 $R/tools/diag/assembly_only.py $rg 3 > $R/gpurun_out/r4_pair_${rg}_$i.log 2>&1
    echo "range $rg pass $i rc=$?"
  done
done
