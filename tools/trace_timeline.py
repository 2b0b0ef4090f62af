#!/usr/bin/env python3
"""Print the kernel timeline of the LAST evaluation in a rocprofv3 kernel-trace CSV (start/end in us
relative to its assembly kernel) -- used to see what overlaps what in the factorisation."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pair_sym" in r["Kernel_Name"]]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
s = idx[which]
e = idx[which + 1] if which + 1 < 0 else len(rows)
t0 = int(rows[s]["Start_Timestamp"])
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 80


def short(n):
    for k in ("pair_sym", "potrf_engine", "potrf_tile", "trsm_tile", "update_kernel<64, 8, 0>",
              "update_kernel<64, 8, 1>", "finalize", "rhs_rows", "loc_params"):
        if k in n:
            return k
    return n[:30]


prev_end = 0.0
for r in rows[s:e][:limit]:
    a = (int(r["Start_Timestamp"]) - t0) / 1e3
    b = (int(r["End_Timestamp"]) - t0) / 1e3
    print("%-26s start %8.1f end %8.1f dur %7.1f gap %6.1f grid %s" % (short(r["Kernel_Name"]), a, b, b - a,
                                                                 a - prev_end, r["Grid_Size_X"]))
    if "engine" not in r["Kernel_Name"]:
        prev_end = b
