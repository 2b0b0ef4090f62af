#!/usr/bin/env python3
"""One batch of independent evaluations at n = 4096 (to be run under rocprofv3 --kernel-trace): how many evaluations really
overlap on the device?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cocons_amd as ca
from cocons_amd import workloads as wl

g = int(sys.argv[1]) if len(sys.argv) > 1 else 64
locs = wl.grid_locs(g)
X = wl.design_from_locs(locs)["std.covs"]
th = wl.theta_full()
fit = ca.CoconsFit(locs, X, wl.synthetic_z(g * g), wl.SMOOTH_LIMITS)
ths = []
for i in range(24):
    t = {k: np.array(v, dtype=np.float64) for k, v in th.items()}
    t["std.dev"][0] += 1.22e-4 * (i + 1)
    ths.append(t)
fit.neg2loglik_batch_core(ths[:8])
fit.neg2loglik_batch_core(ths)
fit.close()
