#!/bin/bash
# One driver for everything that runs on the GPU box (one `gpurun` call each):
#   tools/gpu_run.sh full                 the driver's sequence: whole -m gpu suite, smoke, default bench line
#   tools/gpu_run.sh profile              the round's evidence: bench line, rocprofv3 kernel trace + stats of the bench command, per-task
#                                         trace of the persistent launch, batch probe, sharded schedule on one rank, then `pmc_dag`
#   tools/gpu_run.sh pmc_dag              counters of cocons::dag_kernel replayed alone (one rocprofv3 --pmc pass per group)
#   tools/gpu_run.sh timeline [n] [tag]   kernel timeline of one evaluation (what the main stream does, launch by launch)
#   tools/gpu_run.sh soak [evals]         hand-off time-outs: long runs at several sizes and schedules
#   tools/gpu_run.sh ab "a:k=v" "b:k=v"   schedule variants alternated in one process at n = 10^4 and n = 4096 (tools/ab_modes.py)
#   tools/gpu_run.sh overlap              do the fp64 vector and matrix pipes run side by side; assembly beside the factorisation
# Output goes to gpurun_out/${TAG}_* (TAG defaults to r06); tools/profile_collect.py copies the summaries into profiles/.
R=$PWD
TAG=${TAG:-r06}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
task=${1:-full}; shift

bench_summary() {
python3 - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value", "ms_per_step", "cholesky_frac", "parity_rel_err_vs_cpu")})
print("roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "launch_ms", "traffic", "traffic_algorithmic", "pipe_busy_frac_pmc")})
print("stages", d["stages_ms"])
print("batch", d["throughput_batch_api"], "inflight", d["throughput_inflight"], "taper", d["taper_path"]["ms_per_eval"])
c = d.get("configs") or {}
if c:
    print("C2", c["C2"]["evals_per_s"], c["C2"]["cholesky_frac"], "C4", c["C4"]["sequential"], c["C4"]["batched_gradient_points"], "C5", c["C5"]["wall_ms_min"], c["C5"]["tflops_fp64"])
print("engine", d["engine"])
PY
}

pmc_dag() {
  cd /tmp
  P="python3 $R/tools/dag_replay.py --reps 2 --warm 1"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_dag_trace -o t -- $P > $O/${TAG}_dag_trace.log 2>&1
  echo "replay trace rc=$?"
  i=0
  for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS" \
             "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/${TAG}_dag_pmc_$i -o p -- $P > $O/${TAG}_dag_pmc_$i.log 2>&1
    echo "pmc pass $i ($pmc) rc=$?"
  done
  cd $R
}

case "$task" in
full)
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_gputest_full.log 2>&1; rc=$?; tail -4 $O/${TAG}_gputest_full.log; echo "pytest rc=$rc"
  [ $rc -ne 0 ] && exit $rc
  python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || exit 1
  timeout -k 10 300 python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; rc3=$?; tail -3 $O/${TAG}_bench.err; echo "bench rc=$rc3"
  [ $rc3 -eq 0 ] && bench_summary $O/${TAG}_bench.json
  exit $rc3 ;;
profile)
  python3 bench.py > $O/${TAG}_bench_final.json 2> $O/${TAG}_bench_final.err; echo "bench rc=$?"; bench_summary $O/${TAG}_bench_final.json
  python3 tools/dag_trace.py --n 10000 --every 1 > $O/${TAG}_dag_trace_n10000.txt 2>&1; echo "dag trace rc=$?"
  python3 tools/batch_probe.py > $O/${TAG}_batch_probe.txt 2>&1; echo "batch probe rc=$?"; cat $O/${TAG}_batch_probe.txt
  cd /tmp
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_trace -o t -- python3 $R/bench.py --no-cpu-baseline --no-configs --steps 15 --warmup 2 --inflight 0 > $O/${TAG}_prof_trace.log 2>&1
  echo "bench trace rc=$?"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_shard_trace -o t -- python3 $R/tools/diag/rccl_one_rank.py 100 > $O/${TAG}_shard_trace.log 2>&1
  echo "shard trace rc=$?"
  cd $R
  pmc_dag ;;
pmc_dag) pmc_dag ;;
timeline)
  N=${1:-10000}; T=${2:-tl}
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_$T -o t -- python3 $R/bench.py --n $N --no-cpu-baseline --no-configs --steps 6 --warmup 2 --inflight 0 > $O/${TAG}_$T.log 2>&1
  echo "trace rc=$?"
  cd $R
  f=$(find $O/${TAG}_$T -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_timeline.py $f 400 > $O/${TAG}_timeline_$T.txt 2>&1; tail -5 $O/${TAG}_timeline_$T.txt ;;
soak)
  N=${1:-3000}
  {
    echo "== n=10000 default"; COCONS_DEBUG_ABORT=1 timeout -k 10 300 python3 tools/soak.py $N 2>&1 | tail -4
    echo "== n=4096 default"; COCONS_DEBUG_ABORT=1 timeout -k 10 200 python3 tools/ab_modes.py --n 4096 --rounds 5 --evals 1000 "m0:" 2>&1 | tail -2
    echo "== n=10000 every step under the persistent launch"; COCONS_DEBUG_ABORT=1 COCONS_DAG_MIN_TILES=0 timeout -k 10 200 python3 tools/soak.py 1000 2>&1 | tail -4
    echo "== n=2115 (a last block of one tile)"; COCONS_DEBUG_ABORT=1 timeout -k 10 200 python3 tools/ab_modes.py --n 2116 --rounds 5 --evals 1000 "m0:" 2>&1 | tail -2
  } 2>&1 | tee $O/${TAG}_soak.txt ;;
ab)
  { for n in 10000 4096; do COCONS_DEBUG_ABORT=1 timeout -k 10 400 python3 tools/ab_modes.py --n $n --rounds 7 --evals $([ $n = 4096 ] && echo 60 || echo 15) "$@" 2>&1 | tail -$(( $# + 2 )); done; } | tee $O/${TAG}_ab.txt ;;
overlap)
  python3 tools/diag/corun.py 2>&1 | tee $O/${TAG}_corun.txt
  python3 tools/diag/overlap_probe.py 2>&1 | tee $O/${TAG}_overlap_probe.txt ;;
*) echo "unknown task $task"; exit 2 ;;
esac
