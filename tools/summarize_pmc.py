#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/r3_pmc.sh (gpurun_out/r3_pmc_*/p_counter_collection.csv,
gpurun_out/r3_prof_trace) into the committed profiles/r03_*.json / .csv files.
    python tools/summarize_pmc.py [run-prefix r3] [profile-prefix r03]"""
import csv
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

RUN = sys.argv[1] if len(sys.argv) > 1 else "r3"
OUT = sys.argv[2] if len(sys.argv) > 2 else "r03"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(R, "gpurun_out")
P = os.path.join(R, "profiles")
commit = subprocess.run(["git", "-C", R, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()


def load(i):
    rows = list(csv.DictReader(open(os.path.join(G, "%s_pmc_%d" % (RUN, i), "p_counter_collection.csv"))))
    per = defaultdict(dict)      # dispatch -> {counter: value, name, dur}
    for r in rows:
        d = per[int(r["Dispatch_Id"])]
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["name"] = r["Kernel_Name"]
        d["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        d["grid"] = int(r["Grid_Size"])
    return [per[k] for k in sorted(per)]


def last_eval(disp, key):
    """dispatches of the last complete evaluation (from its pair_sym kernel to the next)"""
    idx = [i for i, d in enumerate(disp) if "pair_sym" in d["name"]]
    s, e = idx[-2], idx[-1]
    return [d for d in disp[s:e] if (key(d["name"]) if callable(key) else key in d["name"])]


def trailing(name):
    """the trailing-update launches: role 0 of update_kernel, 4- or 8-wave instantiation"""
    return "update_kernel<64," in name and (", 0, false" in name or name.rstrip().endswith("<64, 8, 0>") or "<64, 8, 0>" in name)


shutil.copy(os.path.join(G, "%s_prof_trace" % RUN, "t_kernel_stats.csv"), os.path.join(P, "%s_bench_n10000_kernel_stats.csv" % OUT))

# ---- assembly kernel (pair_sym): VALU counters ---------------------------------------------------
a1 = last_eval(load(1), "pair_sym")[0]
a2 = last_eval(load(2), "pair_sym")[0]
pairs = 10000 * 10001 / 2.0
clock = a1["GRBM_GUI_ACTIVE"] / 8.0 / (a1["dur_us"] * 1e-6) / 1e9
out = {
    "kernel": "cocons::pair_sym_kernel<0, false> (general-nu Bessel-K assembly), n = 10000, range 0.05",
    "commit": commit, "command": "tools/%s_pmc.sh (rocprofv3 --pmc, COCONS_ENGINE=0, bench.py --steps 3)" % RUN,
    "duration_us": a1["dur_us"], "pairs": pairs, "pairs_per_s": pairs / (a1["dur_us"] * 1e-6),
    "counters": {k: v for k, v in {**a1, **a2}.items() if k not in ("name", "dur_us", "grid")},
    "valu_wave_instructions_per_pair": a1["SQ_INSTS_VALU"] * 64.0 / pairs / 64.0 * 1.0,
    "note_units": "SQ_INSTS_VALU counts wave-instructions; SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles "
                  "(4 shader cycles) summed over waves; SQ_BUSY_CU_CYCLES per CU",
    "valu_instructions_per_wave_of_64_pairs": a1["SQ_INSTS_VALU"] / (pairs / 64.0),
    "valu_active_frac_of_wave_cycles": a1["SQ_ACTIVE_INST_VALU"] / a1["SQ_WAVE_CYCLES"],
    "wait_inst_any_frac": a2["SQ_WAIT_INST_ANY"] / (a2["SQ_WAIT_INST_ANY"] + a2["SQ_WAIT_ANY"] + a2["SQ_ACTIVE_INST_ANY"]),
    "wait_any_frac": a2["SQ_WAIT_ANY"] / (a2["SQ_WAIT_INST_ANY"] + a2["SQ_WAIT_ANY"] + a2["SQ_ACTIVE_INST_ANY"]),
    "clock_GHz_from_GRBM_GUI_ACTIVE": clock,
    "hbm_write_GBps": 8.0 * pairs / (a1["dur_us"] * 1e-6) / 1e9,
}
json.dump(out, open(os.path.join(P, "%s_pair_sym_valu.json" % OUT), "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])

# ---- update kernel: MFMA busy, LDS, HBM traffic ---------------------------------------------------
u3 = last_eval(load(3), trailing)
fe = last_eval(load(4), trailing)
wr = last_eval(load(5), trailing)
tc = last_eval(load(6), trailing)
tot = lambda L, k: sum(d[k] for d in L)
mf = {
    "kernel": "cocons::update_kernel<64, {8,16}, 0, false, {4,8}> (trailing update), 39 launches of one evaluation at n = 10000",
    "commit": commit, "command": "tools/%s_pmc.sh pass 3" % RUN,
    "mfma_busy_over_simd_cycles_all_launches": tot(u3, "SQ_VALU_MFMA_BUSY_CYCLES") / (tot(u3, "GRBM_GUI_ACTIVE") / 8.0 * 1024.0),
    "definition": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)",
    "mfma_instructions_per_eval": tot(u3, "SQ_INSTS_MFMA"),
    "mfma_busy_cycles_per_instruction": tot(u3, "SQ_VALU_MFMA_BUSY_CYCLES") / max(tot(u3, "SQ_INSTS_MFMA"), 1.0),
    "lds_bank_conflict_frac": tot(u3, "SQ_LDS_BANK_CONFLICT") / max(tot(u3, "SQ_LDS_IDX_ACTIVE"), 1.0),
    "first_launches": [{"dur_us": d["dur_us"], "mfma_busy_frac": d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0),
                        "clock_GHz": d["GRBM_GUI_ACTIVE"] / 8.0 / (d["dur_us"] * 1e-6) / 1e9} for d in u3[:4]],
}
json.dump(mf, open(os.path.join(P, "%s_update_kernel_mfma_util.json" % OUT), "w"), indent=1)
print(json.dumps(mf, indent=1)[:1200])
fetch_kb, write_kb = tot(fe, "FETCH_SIZE"), tot(wr, "WRITE_SIZE")
n_l = len(fe)
alg_c = 5.3e9
tr = {
    "kernel": "cocons::update_kernel<64, {8,16}, 0, false, {4,8}> (trailing launches, K = 256), n = 10000", "commit": commit,
    "launches_per_eval": n_l,
    "FETCH_SIZE_KB_per_eval_raw": fetch_kb, "WRITE_SIZE_KB_per_eval": write_kb,
    "l2_hit_rate": tot(tc, "TCC_HIT_sum") / max(tot(tc, "TCC_HIT_sum") + tot(tc, "TCC_MISS_sum"), 1.0),
    "algorithmic_C_bytes_per_eval": {"read": alg_c, "write": alg_c},
    "hbm_bytes_per_launch_low": (fetch_kb * 1024.0 + write_kb * 1024.0) / n_l,
    "hbm_bytes_per_launch_high": (2 * fetch_kb * 1024.0 + write_kb * 1024.0) / n_l,
    "hbm_bytes_per_launch": (fetch_kb * 1024.0 + write_kb * 1024.0) / n_l,
    "note": "FETCH_SIZE / WRITE_SIZE from separate rocprofv3 --pmc passes.  The microarchitecture guide's x2 correction "
            "of FETCH_SIZE is calibrated for 16 B/lane streaming reads; this kernel's C read-modify-write is 8 B/lane "
            "and its operand reads are 16 B/lane, so the true read volume lies between raw (low) and 2 x raw (high); "
            "`hbm_bytes_per_launch` is the uncorrected sum.  WRITE_SIZE equals the algorithmic C write.",
}
json.dump(tr, open(os.path.join(P, "%s_update_kernel_hbm_traffic.json" % OUT), "w"), indent=1)
print(json.dumps(tr, indent=1))
