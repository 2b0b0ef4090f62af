#!/bin/bash
# engine check: parity suite, then the bench with and without the diagonal-tile engine, then a kernel trace
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2_la_tests.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r2_la_tests.log
tail -3 gpurun_out/r2_la_tests.log
for la in 0 1; do
  COCONS_ENGINE=$la timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r2_la_bench_$la.log 2>&1
  echo "bench la=$la rc=$?"
  python - <<PY
import json
for l in open("gpurun_out/r2_la_bench_$la.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("la=$la value", d["value"], "ms", d["ms_per_step"], "stages", d["stages_ms"], "batch", d["throughput_batch_api"])
PY
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_la_prof -o la -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 --inflight 0 > $GRAFT_REPO_ROOT/gpurun_out/r2_la_prof.log 2>&1
echo "prof rc=$?"
find $GRAFT_REPO_ROOT/gpurun_out/r2_la_prof -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-160
