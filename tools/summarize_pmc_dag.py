#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of `tools/gpu_run.sh pmc_dag` (gpurun_out/<TAG>_dag_pmc_*/p_counter_collection.csv) into
profiles/<TAG>_dag_kernel_mfma_util.json and profiles/<TAG>_dag_kernel_hbm_traffic.json (what bench.py's roofline block reads;
TAG from the environment, default r06).
    python tools/summarize_pmc_dag.py"""
import csv
import glob
import json
import os
import shutil
import subprocess
from collections import defaultdict

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(R, "gpurun_out")
P = os.path.join(R, "profiles")
TAG = os.environ.get("TAG", "r06")
import hashlib
with open(os.path.join(R, "cocons_amd", "csrc", "chol.hip"), "rb") as _fh:
    SRC_SHA = hashlib.sha256(_fh.read()).hexdigest()[:16]      # bench.py compares it with the tree it runs in
commit = subprocess.run(["git", "-C", R, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()


def find(d, name):
    hits = glob.glob(os.path.join(G, d, "**", name), recursive=True)
    if not hits:
        raise SystemExit("missing %s under %s" % (name, d))
    return hits[0]


def load(i):
    rows = list(csv.DictReader(open(find("%s_dag_pmc_%d" % (TAG, i), "p_counter_collection.csv"))))
    per = defaultdict(dict)
    for r in rows:
        d = per[int(r["Dispatch_Id"])]
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["name"] = r["Kernel_Name"]
        d["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return [per[k] for k in sorted(per) if "dag_kernel" in per[k]["name"]]


def last(i):
    d = load(i)
    if not d:
        raise SystemExit("no dag_kernel dispatch in pass %d" % i)
    return d[-1]            # the last replay of the run (warm)


shutil.copy(find(TAG + "_dag_trace", "t_kernel_stats.csv"), os.path.join(P, TAG + "_dag_replay_kernel_stats.csv"))
line = None
for ln in open(os.path.join(G, TAG + "_dag_trace.log")):
    if ln.startswith("{"):
        line = json.loads(ln)
a, b, fe, wr, tc = last(1), last(2), last(3), last(4), last(5)
simd_cycles = a["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
mf = {
    "kernel": "cocons::dag_kernel, replayed alone at n = 10000 (tools/dag_replay.py: same task list, products and C traffic; the "
              "engine's outputs prepared beforehand)",
    "commit": commit, "kernel_source_sha16": SRC_SHA, "command": "tools/gpu_run.sh pmc_dag (rocprofv3 --kernel-trace --pmc, one pass per counter group)",
    "launch_us_under_pmc": a["dur_us"],
    "replay_line_kernel_trace_pass": line,
    "mfma_busy_over_simd_cycles": a["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
    "definition": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)",
    "mfma_instructions": a["SQ_INSTS_MFMA"],
    "mfma_busy_cycles_per_instruction": a["SQ_VALU_MFMA_BUSY_CYCLES"] / max(a["SQ_INSTS_MFMA"], 1.0),
    "clock_GHz_from_GRBM_GUI_ACTIVE": a["GRBM_GUI_ACTIVE"] / 8.0 / (a["dur_us"] * 1e-6) / 1e9,
    "flops_from_mfma_count": a["SQ_INSTS_MFMA"] * 2048.0,
    "wave_cycle_split": {k: b[k] / max(b["SQ_WAIT_INST_ANY"] + b["SQ_WAIT_ANY"] + b["SQ_ACTIVE_INST_ANY"], 1.0)
                         for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY")},
    "counters": {k: v for k, v in {**a, **b}.items() if k not in ("name", "dur_us")},
}
json.dump(mf, open(os.path.join(P, TAG + "_dag_kernel_mfma_util.json"), "w"), indent=1)
print(json.dumps(mf, indent=1)[:1800])
fetch_b, write_b = fe["FETCH_SIZE"] * 1024.0, wr["WRITE_SIZE"] * 1024.0
tr = {
    "kernel": mf["kernel"], "commit": commit, "kernel_source_sha16": SRC_SHA, "command": mf["command"],
    "FETCH_SIZE_bytes_raw": fetch_b, "WRITE_SIZE_bytes": write_b,
    "l2_hit_rate": tc["TCC_HIT_sum"] / max(tc["TCC_HIT_sum"] + tc["TCC_MISS_sum"], 1.0),
    "hbm_bytes_per_launch": fetch_b + write_b,
    "hbm_bytes_per_launch_high": 2.0 * fetch_b + write_b,
    "note": "FETCH_SIZE / WRITE_SIZE from separate rocprofv3 --pmc passes of the replayed launch, per launch.  The guide's x2 "
            "correction of FETCH_SIZE on gfx950 is calibrated for 16 B/lane streaming reads; this kernel reads its C tiles 8 B per "
            "lane (L2-bypassing) and its operand chunks 16 B per lane, so the true read volume lies between raw and 2 x raw "
            "(`hbm_bytes_per_launch_high`); `hbm_bytes_per_launch` is the uncorrected sum.",
}
json.dump(tr, open(os.path.join(P, TAG + "_dag_kernel_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(tr, indent=1))
