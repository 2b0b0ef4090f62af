#!/usr/bin/env python3
"""Hand-off time-outs on FRESH handles in a process that has created and destroyed handles before (the situation of a test
runner or an R session fitting one model after another): per cycle a new handle, a batch (slots = clones with streams of
their own), a few dozen sequential evaluations, close.  Prints retries and abort codes per cycle."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cocons_amd as ca
from cocons_amd import workloads as wl

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 12
total = 0
for c in range(cycles):
    g = 64 if c % 2 == 0 else 100
    locs = wl.grid_locs(g)
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full()
    z = np.column_stack([wl.synthetic_z(g * g), wl.synthetic_z(g * g, seed=3)])
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS, x_betas=X)
    t0 = time.perf_counter()
    vals = [fit.neg2loglik_core(th)[0] for _ in range(30)]
    fit.neg2loglik_profile_core(th)
    ths = []
    for i in range(12):
        t = {k: np.array(v, dtype=np.float64) for k, v in th.items()}
        t["std.dev"][0] += 1.22e-4 * (i + 1)
        ths.append(t)
    fit.neg2loglik_batch_core(ths)
    vals += [fit.neg2loglik_core(th)[0] for _ in range(10)]
    st = fit.engine_state()
    total += st["retries"]
    print("cycle %2d n=%5d: %.0f ms, retries %d, last abort 0x%x, values equal %s"
          % (c, g * g, 1e3 * (time.perf_counter() - t0), st["retries"], st["last_abort"], len(set(vals)) == 1), flush=True)
    fit.close()
print("total retries %d in %d cycles" % (total, cycles))
