#!/bin/bash
# this tree against the build in old_r2_tmp/w_prev on ONE box, alternating (the only cross-build comparison that means anything)
# The reference build is NOT tracked: put a build of the commit to compare with there first, e.g.
#   git worktree add /tmp/ref <commit> && make -C /tmp/ref/cocons_amd/csrc && mkdir -p old_r2_tmp/w_prev &&
#   cp -r /tmp/ref/{bench.py,cocons_amd,include,oracle,tools} old_r2_tmp/w_prev/      (built .so files travel with gpurun)
[ -d old_r2_tmp/w_prev ] || { echo "no reference build under old_r2_tmp/w_prev (see the header of this script)"; exit 2; }
mkdir -p gpurun_out
P=old_r2_tmp/w_prev
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tests_full.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r3_tests_full.log
for rep in 1 2; do
  for w in prev new; do
    if [ $w = prev ]; then D=$P; else D=.; fi
    timeout -k 10 300 python $D/tools/ab_modes.py --rounds 3 --evals 60 "m0:" > gpurun_out/vs_${w}_$rep.log 2>&1
    echo "$w n=10000 $(tail -1 gpurun_out/vs_${w}_$rep.log | cut -c1-120)"
    timeout -k 10 300 python $D/tools/ab_modes.py --n 4096 --rounds 3 --evals 100 "m0:" > gpurun_out/vs4k_${w}_$rep.log 2>&1
    echo "$w n=4096  $(tail -1 gpurun_out/vs4k_${w}_$rep.log | cut -c1-120)"
    timeout -k 10 300 python $D/tools/taper_timing.py 100 0.06 nocpu > gpurun_out/vstaper_${w}_$rep.log 2>&1
    echo "$w taper   $(grep 'taper objective' gpurun_out/vstaper_${w}_$rep.log)"
  done
done
