#!/bin/bash
# final round-3 pass on one box: GPU suite, bench, range sweep, then the rocprofv3 passes behind profiles/r03_*
mkdir -p gpurun_out
rm -rf gpurun_out/r3_pmc_* gpurun_out/r3_prof_trace
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tests_full.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r3_tests_full.log
timeout -k 10 400 python bench.py > gpurun_out/r3_bench_full.log 2>&1
echo "bench rc=$?"; tail -1 gpurun_out/r3_bench_full.log | cut -c1-400
timeout -k 10 300 python tools/range_sweep.py > gpurun_out/r3_range_sweep.log 2>&1
echo "range rc=$?"; cat gpurun_out/r3_range_sweep.log
bash tools/r3_pmc.sh
# keep the merged output small: only the CSVs the summariser reads
find gpurun_out/r3_pmc_* gpurun_out/r3_prof_trace -type f ! -name "*counter_collection.csv" ! -name "*kernel_stats.csv" ! -name "*kernel_trace.csv" -delete 2>/dev/null
du -sh gpurun_out | tail -1
