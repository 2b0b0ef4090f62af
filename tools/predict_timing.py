#!/usr/bin/env python3
"""BASELINE config C5 on one GPU: cocoPredict core, n_train = m_pred = 8192 (128 x 64 grid and the
half-cell-shifted grid), wall time of cocons_predict_dense (host vectors out)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocons_amd as ca
from cocons_amd import workloads as wl

locs = wl.grid_locs(128, 64)
sc = wl.design_from_locs(locs)
X = sc["std.covs"]
th = wl.theta_full()
th["mean"] = np.array([0.3, -0.1, 0.2])
z = wl.synthetic_z(8192)
lp = locs + np.array([0.5 / 127, 0.5 / 63])
Xp = wl.design_from_locs(lp, sc["mean.vector"], sc["sd.vector"])["std.covs"]
fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
fit.predict_core(th, lp, Xp)
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    st, qf = fit.predict_core(th, lp, Xp)
    ts.append(time.perf_counter() - t0)
n = m = 8192
flops = n ** 3 / 3 + m * n * n          # factorisation + bordered rows (solve of m right-hand sides)
print("C5 predict n=%d m=%d: %.2f ms (min of 5; median %.2f) -> %.1f TFLOP/s on n^3/3 + m n^2 flop" %
      (n, m, 1e3 * min(ts), 1e3 * sorted(ts)[2], flops / min(ts) / 1e12))
