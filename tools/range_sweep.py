#!/usr/bin/env python3
"""Assembly / evaluation time against the correlation range (n = 10 000): longer ranges put more
pairs at small u, where K_nu needs Temme's series or many CF2 steps."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocons_amd as ca
from cocons_amd import workloads as wl

g = int(sys.argv[1]) if len(sys.argv) > 1 else 100
locs = wl.grid_locs(g)
X = wl.design_from_locs(locs)["std.covs"]
z = wl.synthetic_z(g * g)
fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
for rng_ in (0.02, 0.05, 0.1, 0.2, 0.5, 1.0):
    th = wl.theta_full(scale0=np.log(rng_))
    try:
        st = fit.profile_stages(th, reps=2)
        print("range %.2f: assembly %.2f ms, cholesky %.2f ms, eval %.2f ms" %
              (rng_, st["assembly_ms"], st["cholesky_ms"], st["eval_ms"]))
    except Exception as e:
        print("range %.2f: %s" % (rng_, e))
