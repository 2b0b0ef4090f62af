#!/usr/bin/env python3
"""Copy the summaries of `tools/gpu_run.sh profile` (and of its other tasks, where present) from gpurun_out/ (scratch) into
profiles/ (tracked), named per round: TAG from the environment, default r06."""
import glob
import os
import shutil
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
TAG = os.environ.get("TAG", "r06")


def find(d, name):
    hits = glob.glob(os.path.join(G, d, "**", name), recursive=True)
    return hits[0] if hits else None


for src, dst in (("bench_final.json", "bench_n10000.json"), ("dag_trace_n10000.txt", "dag_trace_n10000.txt"),
                 ("batch_probe.txt", "batch_probe.txt"), ("soak.txt", "soak.txt"), ("ab.txt", "ab.txt"),
                 ("corun.txt", "corun_probe.txt"), ("overlap_probe.txt", "overlap_probe.txt"),
                 ("timeline_tl.txt", "timeline_n10000.txt"), ("timeline_tl4096.txt", "timeline_n4096.txt")):
    if os.path.exists(os.path.join(G, TAG + "_" + src)):
        shutil.copy(os.path.join(G, TAG + "_" + src), os.path.join(P, TAG + "_" + dst))
        print("copied", TAG + "_" + dst)
for d, dst in (("prof_trace", "bench_n10000_kernel_stats.csv"), ("shard_trace", "shard_one_rank_kernel_stats.csv")):
    f = find(TAG + "_" + d, "t_kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(P, TAG + "_" + dst))
        print("copied", TAG + "_" + dst)
if glob.glob(os.path.join(G, TAG + "_dag_pmc_1")):
    subprocess.call([sys.executable, os.path.join(R, "tools", "summarize_pmc_dag.py")])
