#!/usr/bin/env python3
"""(scratch) stamps of the follow layout's followers against the engine's, late steps"""
import ctypes, math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cocons_amd as ca
from cocons_amd import _lib, workloads as wl
L = _lib.load()
for k, v in (("dag", 1), ("dag_chain", 2), ("dag_min_tiles", 0), ("dag_trace", 1)):
    _lib.check(L.cocons_debug_tune(k.encode(), v), "tune")
n = 10000
g = int(round(math.sqrt(n)))
locs = wl.grid_locs(g); X = wl.design_from_locs(locs)["std.covs"]; th = wl.theta_full()
fit = ca.CoconsFit(locs, X, wl.synthetic_z(g * g), wl.SMOOTH_LIMITS)
for _ in range(3):
    fit.neg2loglik_core(th)
ns = ctypes.c_int(0)
nt_tasks = L.cocons_debug_dag_trace(fit._h, ctypes.byref(ns), None, None, None)
steps = np.zeros((ns.value, 20), dtype=np.int32); stamps = np.zeros((nt_tasks, 4), dtype=np.uint64)
nt = (fit.n + 127) // 128 + 2
eng = np.zeros((nt + 2, 8), dtype=np.uint64)
L.cocons_debug_dag_trace(fit._h, ctypes.byref(ns), steps.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                         stamps.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), eng.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)))
nc = L.cocons_debug_chain_trace(fit._h, None)
cst = np.zeros((max(nc, 1), 4), dtype=np.uint64)
L.cocons_debug_chain_trace(fit._h, cst.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)))
print("steps", ns.value, "chain entries", nc, fit.engine_state())
for s in list(range(12, 16)) + list(range(28, 37)):
    e = eng[s + 1].astype(np.int64); nx = eng[s + 2].astype(np.int64)
    if e[0] == 0: continue
    t0 = e[0]
    us = lambda v: (int(v) - t0) * 0.01 if v else float("nan")
    print("step %2d engine: tile t %.1f out[t] %.1f xr %.1f tile t+1 %.1f out[t+1] %.1f | next in seen %.1f" % (s, us(e[1]), us(e[2]), us(e[4]), us(e[6]), us(e[7]), us(nx[0])))
    for r in range(1, 8):
        c = cst[s * 8 + r].astype(np.int64)
        print("     role %d (%s): %s" % (r, "strips" if r <= 2 else "diag", "  ".join("%7.1f" % us(x) for x in c)))
