#!/bin/bash
# bench variants selected by environment: usage tools/r2_var.sh "VAR=1 VAR2=0" ...
mkdir -p gpurun_out
i=0
for v in "$@"; do
  i=$((i+1))
  env $v timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --inflight 0 > gpurun_out/r2_var_$i.log 2>&1
  python - "$v" gpurun_out/r2_var_$i.log <<'PY'
import json, sys
for l in open(sys.argv[2]):
    if l.startswith("{"):
        d = json.loads(l)
        st = d["stages_ms"]
        print("%-40s value %7.2f ms/step %7.3f asm %.3f chol %.3f upd_sum %.3f (%d launches)" % (sys.argv[1], d["value"], d["ms_per_step"], st["assembly_ms"], st["cholesky_ms"], st["update_sum_ms"], st["update_launches"]))
PY
done
