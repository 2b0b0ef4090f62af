#!/usr/bin/env python3
"""Throughput of cocons_neg2loglik_batch at n = 10 000 for a batch of 33 points (1 + 2P, P = 16)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocons_amd as ca
from cocons_amd import workloads as wl

g = int(sys.argv[1]) if len(sys.argv) > 1 else 100
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 33
locs = wl.grid_locs(g)
X = wl.design_from_locs(locs)["std.covs"]
th = wl.theta_full()
z = wl.synthetic_z(g * g)
fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
ths = []
for i in range(nb):
    t = {k: np.array(v, dtype=np.float64) for k, v in th.items()}
    t["std.dev"][0] += 1.22e-4 * (i + 1)
    ths.append(t)
fit.neg2loglik_batch_core(ths[:6])
t0 = time.perf_counter()
v, s = fit.neg2loglik_batch_core(ths)
dt = time.perf_counter() - t0
print("slots=%s n=%d batch=%d: %.2f evals/s (%.2f ms/eval) ok=%s" %
      (os.environ.get("COCONS_BATCH_SLOTS", "3"), g * g, nb, nb / dt, 1e3 * dt / nb, bool((s == 0).all())))
