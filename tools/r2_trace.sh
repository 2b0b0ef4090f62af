#!/bin/bash
# kernel traces of bench.py under environment variants: tools/r2_trace.sh tag "ENV=.." [tag "ENV=.."]...
R=$PWD
export TMPDIR=/tmp
while [ $# -ge 2 ]; do
  tag=$1; v=$2; shift 2
  cd /tmp
  export $v
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_tr_$tag -o t -- python3 $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 --inflight 0 > $R/gpurun_out/r2_tr_$tag.log 2>&1
  echo "$tag rc=$?"
  for n in $(echo $v | tr ' ' '\n' | cut -d= -f1); do unset $n; done
  cd $R
done
