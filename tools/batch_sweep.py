#!/usr/bin/env python3
"""Throughput of cocons_neg2loglik_batch at n = 10 000 (environment selects slots / engine use)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cocons_amd as ca
from cocons_amd import workloads as wl

g = 100
locs = wl.grid_locs(g)
X = wl.design_from_locs(locs)["std.covs"]
th = wl.theta_full()
fit = ca.CoconsFit(locs, X, wl.synthetic_z(g * g), wl.SMOOTH_LIMITS)
ths = []
for i in range(33):
    t = {k: np.array(v, dtype=np.float64) for k, v in th.items()}
    t["std.dev"][0] += 1.22e-4 * (i + 1)
    ths.append(t)
fit.neg2loglik_batch_core(ths[:6])
t0 = time.perf_counter()
v, s = fit.neg2loglik_batch_core(ths)
dt = time.perf_counter() - t0
print("%-60s batch of 33: %.1f evals/s, all ok %s" % (" ".join("%s=%s" % kv for kv in sorted(os.environ.items()) if kv[0].startswith("COCONS_")),
                                                    33 / dt, bool((s == 0).all())))
