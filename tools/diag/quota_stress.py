#!/usr/bin/env python3
"""An XCD that contributes HALF its share of workgroups to the persistent launch (quota 128 of 255 on the engine's XCD; every step
under the launch, n = 5184): the other XCDs must carry its class of list positions -- no hand-off time-out in `evals` evaluations.
(Below that the library keeps the one counter of rounds 4-5: with a quota of 8 the XCD-aware order crawled at the pace of those
eight workgroups whenever everybody else waited at their tasks, and about one evaluation in 500 ran into a bounded wait.)
    python tools/diag/quota_stress.py [evals = 300] [quota = 128]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cocons_amd as ca
from cocons_amd import _lib, workloads as wl
L = _lib.load()
evals = int(sys.argv[1]) if len(sys.argv) > 1 else 300
g = 72
locs = wl.grid_locs(g); X = wl.design_from_locs(locs)["std.covs"]; th = wl.theta_full()
fit = ca.CoconsFit(locs, X, wl.synthetic_z(g * g), wl.SMOOTH_LIMITS)
quota = int(sys.argv[2]) if len(sys.argv) > 2 else 128
for k, v in (("dag", 1), ("dag_min_tiles", 0), ("dag_xcd_min_quota", 1), ("dag_xcc_quota", quota)):
    _lib.check(L.cocons_debug_tune(k.encode(), v), "tune")
v0 = fit.neg2loglik_core(th)[0]
import time
t0 = time.perf_counter()
for i in range(evals):
    assert fit.neg2loglik_core(th)[0] == v0
dt = time.perf_counter() - t0
print("quota %d, n = %d, %d evaluations: %.2f ms each, engine %s" % (quota, g * g, evals, 1e3 * dt / evals, fit.engine_state()))
fit.close()
