#!/usr/bin/env python3
"""In-kernel clock of the trailing-update kernels during real evaluations at n = 10 000
(COCONS_UPD_STAMP=1 must be set; other COCONS_* knobs select the variant)."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cocons_amd as ca
from cocons_amd import _lib, workloads as wl

L = _lib.load()
L.cocons_debug_upd_clock.restype = ctypes.c_int
L.cocons_debug_upd_clock.argtypes = [ctypes.POINTER(ctypes.c_double)]
g = 100
locs = wl.grid_locs(g)
X = wl.design_from_locs(locs)["std.covs"]
th = wl.theta_full()
fit = ca.CoconsFit(locs, X, wl.synthetic_z(g * g), wl.SMOOTH_LIMITS)
for _ in range(3):
    fit.neg2loglik_core(th)
out = (ctypes.c_double * 3)()
_lib.check(L.cocons_debug_upd_clock(out), "clock")
t0 = time.perf_counter()
n = 10
for _ in range(n):
    fit.neg2loglik_core(th)
dt = (time.perf_counter() - t0) / n
_lib.check(L.cocons_debug_upd_clock(out), "clock")
print("%-50s ms/eval %.3f  update kernels: clock %.3f GHz, %.0f cycles per workgroup, %d workgroups/eval" % (
    " ".join("%s=%s" % kv for kv in sorted(os.environ.items()) if kv[0].startswith("COCONS_") and kv[0] != "COCONS_UPD_STAMP"),
    dt * 1e3, out[0], out[1], out[2] / n))
