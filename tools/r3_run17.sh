#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3_tests_p.log 2>&1
echo "parity tests rc=$?"; tail -3 gpurun_out/r3_tests_p.log
one() {  # dir tag n
  (cd $1 && timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 5 --inflight 0 --n $3 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('evals/s %.2f asm %.3f chol %.3f updsum %.3f' % (d['value'], s['assembly_ms'], s['cholesky_ms'], s['update_sum_ms']))") | sed "s/^/$2 n=$3: /"
}
for rep in 1 2; do
  one old_r2_tmp/w_prev prev 10000
  one . new 10000
done
(cd old_r2_tmp/w_prev && python tools/range_sweep.py 2>&1 | tail -8) | sed 's/^/prev: /'
python tools/range_sweep.py 2>&1 | tail -8 | sed 's/^/new: /'
