#!/bin/bash
# the driver's sequence on one box: whole -m gpu suite, smoke, default bench line
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r5_gputest_full.log 2>&1; rc=$?; tail -4 gpurun_out/r5_gputest_full.log; echo "pytest rc=$rc"
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 300 python3 bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err; rc3=$?; tail -3 gpurun_out/r5_bench.err; echo "bench rc=$rc3"; python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r5_bench.json"))
print({k: d[k] for k in ("value", "ms_per_step", "cholesky_frac", "parity_rel_err_vs_cpu")})
print("roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "traffic", "traffic_algorithmic", "pipe_busy_frac_pmc")})
print("batch", d["throughput_batch_api"], "inflight", d["throughput_inflight"], "taper", d["taper_path"]["ms_per_eval"])
c = d["configs"]
print("C2", c["C2"]["evals_per_s"], c["C2"]["cholesky_frac"], "C4", c["C4"]["sequential"], c["C4"]["batched_gradient_points"], c["C4"]["split_ms_per_eval"], "C5", c["C5"]["wall_ms_min"], c["C5"]["tflops_fp64"])
PY
exit $rc
