#!/usr/bin/env python3
"""A few evaluations at ONE correlation range (n = 10^4) -- the program behind the assembly kernel's PMC passes
(tools/r4_pmc_pair.sh): python tools/diag/assembly_only.py <range> [evaluations]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cocons_amd as ca                     # noqa: E402
from cocons_amd import workloads as wl     # noqa: E402

rng_ = float(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = 100
locs = wl.grid_locs(g)
X = wl.design_from_locs(locs)["std.covs"]
z = wl.synthetic_z(g * g)
fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
th = wl.theta_full(scale0=np.log(rng_))
for _ in range(reps):
    v = fit.neg2loglik_core(th)[0]
print("range %g: -2 loglik %.6f" % (rng_, v))
fit.close()
