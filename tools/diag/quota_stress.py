#!/usr/bin/env python3
"""An XCD that contributes almost no workgroups to the persistent launch (quota 8 of 255 on the engine's XCD; every step under the
launch, n = 5184): the other XCDs must carry its class of list positions -- no hand-off time-out in `evals` evaluations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cocons_amd as ca
from cocons_amd import _lib, workloads as wl
L = _lib.load()
evals = int(sys.argv[1]) if len(sys.argv) > 1 else 300
g = 72
locs = wl.grid_locs(g); X = wl.design_from_locs(locs)["std.covs"]; th = wl.theta_full()
fit = ca.CoconsFit(locs, X, wl.synthetic_z(g * g), wl.SMOOTH_LIMITS)
for k, v in (("dag", 1), ("dag_min_tiles", 0), ("dag_xcc_quota", 8)):
    _lib.check(L.cocons_debug_tune(k.encode(), v), "tune")
v0 = fit.neg2loglik_core(th)[0]
import time
t0 = time.perf_counter()
for i in range(evals):
    assert fit.neg2loglik_core(th)[0] == v0
dt = time.perf_counter() - t0
print("quota 8, n = %d, %d evaluations: %.2f ms each, engine %s" % (g * g, evals, 1e3 * dt / evals, fit.engine_state()))
fit.close()
