#!/bin/bash
# same-box comparison of the round-2 library (old_r2_tmp/) with the current one
mkdir -p gpurun_out
for rep in 1 2; do
  (cd old_r2_tmp && timeout -k 10 200 python tools/taper_timing.py 100 0.06 nocpu 2>&1 | grep -E "taper objective|batch") | sed "s/^/old$rep: /"
  timeout -k 10 200 python tools/taper_timing.py 100 0.06 nocpu 2>&1 | grep -E "taper objective|batch" | sed "s/^/new$rep: /"
  (cd old_r2_tmp && timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --warmup 5 --inflight 0 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('evals/s %.2f asm %.3f chol %.3f updsum %.3f' % (d['value'], s['assembly_ms'], s['cholesky_ms'], s['update_sum_ms']))") | sed "s/^/old$rep bench: /"
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --warmup 5 --inflight 0 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('evals/s %.2f asm %.3f chol %.3f updsum %.3f' % (d['value'], s['assembly_ms'], s['cholesky_ms'], s['update_sum_ms']))" | sed "s/^/new$rep bench: /"
done
