#!/usr/bin/env python3
"""First evaluations under the chain layout, step by step, with a stack dump if anything stands still."""
import faulthandler
import os
import sys
import time

faulthandler.dump_traceback_later(25, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cocons_amd as ca
from cocons_amd import _lib, workloads as wl

L = _lib.load()
g = int(sys.argv[1]) if len(sys.argv) > 1 else 40
mt0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
chain = int(sys.argv[3]) if len(sys.argv) > 3 else 1
print("grid", g, "min_tiles", mt0, "chain", chain, flush=True)
_lib.check(L.cocons_debug_tune(b"dag_min_tiles", mt0), "tune")
_lib.check(L.cocons_debug_tune(b"dag_chain", chain), "tune")
locs = wl.grid_locs(g)
X = wl.design_from_locs(locs)["std.covs"]
th = wl.theta_full()
fit = ca.CoconsFit(locs, X, wl.synthetic_z(g * g), wl.SMOOTH_LIMITS)
print("handle created", flush=True)
_lib.check(L.cocons_debug_tune(b"dag", 0), "tune")
ref = fit.neg2loglik_core(th)[0]
print("classic value", ref, fit.engine_state(), flush=True)
_lib.check(L.cocons_debug_tune(b"dag", 1), "tune")
for i in range(3):
    t0 = time.perf_counter()
    v = fit.neg2loglik_core(th)[0]
    print("dag eval %d: %.3f ms value %r rel %.2e %s" % (i, 1e3 * (time.perf_counter() - t0), v, abs(v - ref) / abs(ref), fit.engine_state()), flush=True)
fit.close()
print("done", flush=True)
