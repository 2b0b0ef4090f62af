#!/usr/bin/env python3
"""Who took part in the persistent launch: workgroups per XCD and the XCDs' task counters after one evaluation at n = 10^4."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cocons_amd as ca
from cocons_amd import _lib, workloads as wl
L = _lib.load()
g = 100
locs = wl.grid_locs(g); X = wl.design_from_locs(locs)["std.covs"]; th = wl.theta_full()
fit = ca.CoconsFit(locs, X, wl.synthetic_z(g * g), wl.SMOOTH_LIMITS)
for xcd in (0, 1):
    _lib.check(L.cocons_debug_tune(b"dag_xcd", xcd), "tune")
    for _ in range(2):
        fit.neg2loglik_core(th)
    w = np.zeros(48, dtype=np.uint32)
    _lib.check(L.cocons_debug_dag_words(fit._h, 48, w.ctypes.data_as(ctypes.POINTER(ctypes.c_uint))), "words")
    print("dag_xcd=%d: counter %d; arrivals counted per XCD %s" % (xcd, w[0], w[16:24].tolist()))
fit.close()
