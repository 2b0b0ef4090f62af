#!/bin/bash
one() {  # dir tag
  (cd $1 && timeout -k 10 200 python tools/taper_timing.py 100 0.06 nocpu 2>&1 | grep -E "taper objective|batch" | tr '\n' ' ') | sed "s/^/$2: /"; echo
  (cd $1 && timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --warmup 5 --inflight 0 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('evals/s %.2f asm %.3f chol %.3f updsum %.3f' % (d['value'], s['assembly_ms'], s['cholesky_ms'], s['update_sum_ms']))") | sed "s/^/$2 bench: /"
}
for rep in 1 2; do
  one old_r2_tmp/w_bfe8a35 c1
  one . new
  COCONS_ENGINE_FUSED=0 one . newf0
done
