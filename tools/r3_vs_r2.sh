#!/bin/bash
# the round-2 build (old_r2_tmp/, not tracked) and this tree on ONE box, alternating: python bench.py as the driver runs it
# The reference build is NOT tracked: put the round-2 tree (tag of the round-2 VERDICT commit) with its library built there first, e.g.
#   git worktree add /tmp/ref <commit> && make -C /tmp/ref/cocons_amd/csrc && mkdir -p old_r2_tmp &&
#   cp -r /tmp/ref/{bench.py,cocons_amd,include,oracle,tools} old_r2_tmp/      (built .so files travel with gpurun)
[ -d old_r2_tmp ] || { echo "no reference build under old_r2_tmp (see the header of this script)"; exit 2; }
mkdir -p gpurun_out
for rep in 1 2 3; do
  for w in r2 r3; do
    if [ $w = r2 ]; then B=old_r2_tmp/bench.py; else B=bench.py; fi
    timeout -k 10 300 python $B --no-cpu-baseline --inflight 0 > gpurun_out/vsr2_${w}_$rep.log 2>&1
    echo "$w rep $rep rc=$? $(python3 - gpurun_out/vsr2_${w}_$rep.log <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = d["stages_ms"]
    print("evals/s %.2f ms %.3f asm %.3f chol %.3f updsum %.3f frac %.4f" % (d["value"], d["ms_per_step"], s["assembly_ms"], s["cholesky_ms"], s["update_sum_ms"], d["roofline"]["frac"]))
except Exception as e:
    print("parse failed", e)
PY
)"
  done
done
