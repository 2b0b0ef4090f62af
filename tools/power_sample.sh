#!/bin/bash
# Sample board power, clocks and temperature (rocm-smi, read-only) while sequential evaluations run:
#   tools/power_sample.sh [evals] > gpurun_out/power_sample.txt
# Evidence for DESIGN section 8: the dense evaluation runs at the board's power limit.
EV=${1:-1500}
OUT=${2:-gpurun_out/power_sample}
mkdir -p gpurun_out
rocm-smi --showmaxpower --showpower --showclocks --showtemp > $OUT.idle.txt 2>&1
python tools/ab_modes.py --rounds 1 --evals $EV "m0:" > $OUT.run.log 2>&1 &
PID=$!
sleep 6        # library load, handle creation, warm-up
: > $OUT.txt
for i in $(seq 1 14); do
  if ! kill -0 $PID 2>/dev/null; then break; fi
  echo "--- sample $i $(date +%s.%N)" >> $OUT.txt
  rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (edge|junction|memory|hotspot)" >> $OUT.txt
  sleep 1
done
wait $PID
echo "run rc=$?"
tail -2 $OUT.run.log
