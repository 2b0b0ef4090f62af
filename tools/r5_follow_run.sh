#!/bin/bash
# potrf_follow_kernel: correctness, then A/B on the paths that use it (taper, plain schedule batch), and that the engine paths do not move
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_engine_sizes.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_dag.py -x -q -m gpu > gpurun_out/follow_tests.log 2>&1
rc=$?; tail -4 gpurun_out/follow_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
{
for fo in 1 0 1 0; do
  echo "== COCONS_POTRF_FOLLOW=$fo"
  COCONS_POTRF_FOLLOW=$fo python tools/taper_timing.py 2>&1 | grep -E "taper objective|batch of"
  COCONS_POTRF_FOLLOW=$fo python tools/batch_probe.py 2>&1 | tail -2
done
python tools/ab_modes.py --n 4096 --rounds 5 --evals 30 "follow:potrf_follow=1" "two:potrf_follow=0" "plainF:engine=0,potrf_follow=1" "plain2:engine=0,potrf_follow=0" 2>&1 | tail -5
} 2>&1 | tee gpurun_out/r5_follow_ab.txt
