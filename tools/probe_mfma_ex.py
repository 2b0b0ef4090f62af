#!/usr/bin/env python3
"""fp64 MFMA ceiling of this GPU: issue rate per clock vs the clock held under load, and the effect of
the duty cycle (bursts with idle gaps) -- cocons_mfma_f64_probe_ex.  Writes a JSON summary to argv[1]."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cocons_amd import _lib

L = _lib.load_probes()          # libcocons_hip_probes.so: the probes are not in the product library
rows = []


def run(bpc, nacc, form, iters, gap_us, reps):
    out = (ctypes.c_double * 4)()
    _lib.check(L.cocons_mfma_f64_probe_ex(bpc, nacc, form, iters, gap_us, reps, out), "probe_ex")
    r = {"blocks_per_cu": bpc, "nacc": nacc, "form": "16x16x4" if form == 0 else "4x4x4_4b", "iters": iters,
         "gap_us": gap_us, "tflops": round(out[0], 2), "clock_ghz": round(out[1], 3),
         "cycles_per_mfma": round(out[2], 2), "burst_ms": round(out[3], 4)}
    rows.append(r)
    print(r)
    sys.stdout.flush()


# issue rate: accumulators per wave x waves per SIMD, long bursts back to back
for form in (0, 1):
    for nacc in (4, 8, 16):
        for bpc in (1, 2, 4, 8):
            run(bpc, nacc, form, 20000 // nacc * 4, 0, 3)
# duty cycle: ~0.3 ms bursts at 8 waves/SIMD with idle gaps in between
for gap in (0, 50, 100, 200, 400, 1000):
    run(8, 4, 0, 1200, gap, 30)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
