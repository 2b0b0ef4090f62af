#!/bin/bash
# A/B of schedule switches on one box: tools/r3_ab.sh tag "ENV=.." [tag "ENV=.."]...   (bench only, short)
R=$PWD
mkdir -p gpurun_out
while [ $# -ge 2 ]; do
  tag=$1; v=$2; shift 2
  export $v
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 --inflight 0 > gpurun_out/r3_ab_$tag.log 2>&1
  echo "$tag rc=$? $(python3 - gpurun_out/r3_ab_$tag.log <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = d["stages_ms"]
    print("evals/s %.2f ms %.3f asm %.3f chol %.3f updsum %.3f frac %.4f" % (d["value"], d["ms_per_step"], s["assembly_ms"], s["cholesky_ms"], s["update_sum_ms"], d["roofline"]["frac"]))
except Exception as e:
    print("parse failed", e)
PY
)"
  for n in $(echo $v | tr ' ' '\n' | cut -d= -f1); do unset $n; done
done
