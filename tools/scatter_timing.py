#!/usr/bin/env python3
"""Assembly time on scattered (randomly ordered) locations, n = 10 000: the spatial sort at fit creation
(COCONS_SPATIAL_SORT) restores the coherence the 8 x 8 pair patches exploit."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocons_amd as ca
from cocons_amd import workloads as wl

rng = np.random.default_rng(1)
n = 10000
locs = rng.uniform(0, 1, size=(n, 2))
X = wl.design_from_locs(locs)["std.covs"]
z = rng.standard_normal(n)
th = wl.theta_full()
fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
st = fit.profile_stages(th, reps=3)
v, _ = fit.neg2loglik_core(th)
print("COCONS_SPATIAL_SORT=%s: assembly %.2f ms, eval %.2f ms, -2loglik %.10f" %
      (os.environ.get("COCONS_SPATIAL_SORT", "1"), st["assembly_ms"], st["eval_ms"], v))
