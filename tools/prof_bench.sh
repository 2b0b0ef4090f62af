#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_bench.sh <tag> [bench args...]
# rocprofv3 kernel trace + stats of bench.py; summaries land in gpurun_out/prof_<tag>/
tag=$1; shift
R=$PWD
export TMPDIR=/tmp
cd /tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/prof_$tag.log 2>&1
cd $R
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls=%5s total_ms=%9.3f avg_us=%9.2f pct=%s" % (r["Name"][:60], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
