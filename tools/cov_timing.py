#!/usr/bin/env python3
"""End-to-end timing of the stateless .Call-surface entry points at n = 10 000 (host buffers in,
host matrix out: PCIe-inclusive) for the closed-form and the Bessel branches."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocons_amd as ca
from cocons_amd import workloads as wl

g = int(sys.argv[1]) if len(sys.argv) > 1 else 100
locs = wl.grid_locs(g)
X = wl.design_from_locs(locs)["std.covs"]
th = wl.theta_full()
ca.cov_rns(th, locs[:64], X[:64], wl.SMOOTH_LIMITS)
for name, lim, sm in (("nu=0.5 closed form", (0.5, 0.5), np.zeros(3)), ("nu=1.5 closed form", (1.5, 1.5), np.zeros(3)),
                      ("general nu (Bessel-K)", wl.SMOOTH_LIMITS, th["smooth"])):
    t = dict(th)
    t["smooth"] = sm
    t0 = time.perf_counter()
    S = ca.cov_rns(t, locs, X, lim)
    dt = time.perf_counter() - t0
    print("cov_rns n=%d %-24s %.1f ms end to end (%.2f GB result over PCIe)" % (g * g, name, 1e3 * dt, S.nbytes / 1e9))
