#!/usr/bin/env python3
"""Copy the summaries of tools/r5_profile.sh from gpurun_out/ (scratch) into profiles/ (tracked)."""
import glob
import os
import shutil
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")


def find(d, name):
    hits = glob.glob(os.path.join(G, d, "**", name), recursive=True)
    return hits[0] if hits else None


for src, dst in (("r5_bench_final.json", "r05_bench_n10000.json"), ("r5_dag_trace_n10000.txt", "r05_dag_trace_n10000.txt"),
                 ("r5_chain_trace_n10000.txt", "r05_chain_trace_n10000.txt"), ("r5_batch_trace_n4096.txt", "r05_batch_trace_n4096.txt"),
                 ("r5_batch_probe_final.txt", "r05_batch_probe.txt")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))
        print("copied", dst)
for d, dst in (("r5_prof_trace", "r05_bench_n10000_kernel_stats.csv"), ("r5_shard_trace", "r05_shard_one_rank_kernel_stats.csv")):
    f = find(d, "t_kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(P, dst))
        print("copied", dst)
subprocess.call([sys.executable, os.path.join(R, "tools", "summarize_pmc_dag.py")])
