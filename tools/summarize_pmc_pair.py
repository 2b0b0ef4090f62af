#!/usr/bin/env python3
"""gpurun_out/r4_pair_<range>_<pass>/p_counter_collection.csv (tools/r4_pmc_pair.sh) -> profiles/r04_pair_sym_valu.json"""
import csv
import json
import os
import subprocess
from collections import defaultdict

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(R, "gpurun_out")
commit = subprocess.run(["git", "-C", R, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
pairs = 10000 * 10001 / 2.0
out = {"kernel": "cocons::pair_sym_kernel<0, false> (general-nu Bessel-K assembly), n = 10000", "commit": commit,
       "command": "tools/r4_pmc_pair.sh (rocprofv3 --pmc, COCONS_ENGINE=0, tools/diag/assembly_only.py <range> 3: last launch)",
       "note_units": "SQ_INSTS_VALU counts wave-instructions; SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles "
                     "(4 shader cycles) summed over waves", "ranges": {}}
for rg in ("0.05", "1.0"):
    c = {}
    dur = None
    for i in (1, 2):
        rows = list(csv.DictReader(open(os.path.join(G, "r4_pair_%s_%d" % (rg, i), "p_counter_collection.csv"))))
        per = defaultdict(dict)
        for r in rows:
            if "pair_sym" not in r["Kernel_Name"]:
                continue
            d = per[int(r["Dispatch_Id"])]
            d[r["Counter_Name"]] = float(r["Counter_Value"])
            d["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        last = per[max(per)]
        if i == 1:
            dur = last["dur_us"]
        c.update({k: v for k, v in last.items() if k != "dur_us"})
    out["ranges"][rg] = {
        "duration_us_under_pmc": dur, "pairs_per_s": pairs / (dur * 1e-6), "counters": c,
        "valu_instructions_per_wave_of_64_pairs": c["SQ_INSTS_VALU"] / (pairs / 64.0),
        "valu_active_frac_of_wave_cycles": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
        "wait_inst_any_frac": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
        "clock_GHz_from_GRBM_GUI_ACTIVE": c["GRBM_GUI_ACTIVE"] / 8.0 / (dur * 1e-6) / 1e9,
    }
json.dump(out, open(os.path.join(R, "profiles", "r04_pair_sym_valu.json"), "w"), indent=1)
print(json.dumps({k: {a: b for a, b in v.items() if a != "counters"} for k, v in out["ranges"].items()}, indent=1))
