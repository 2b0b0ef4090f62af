#!/bin/bash
# this tree's build against tools/diag/prev/libcocons_hip.so (a build of the commit to compare with) on ONE box, alternating
# processes (COCONS_HIP_LIB selects the library): the only cross-build comparison that means anything on this pool
N=${1:-10000}; REPS=${2:-3}; EV=${3:-60}
for rep in $(seq $REPS); do
  for w in prev new; do
    if [ $w = prev ]; then export COCONS_HIP_LIB=$PWD/tools/diag/prev/libcocons_hip.so; else unset COCONS_HIP_LIB; fi
    echo "$w $(timeout -k 10 300 python tools/ab_modes.py --n $N --rounds 3 --evals $EV "m0:" 2>&1 | tail -1 | cut -c1-140)"
  done
done
