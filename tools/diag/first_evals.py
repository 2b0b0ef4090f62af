"""Time the first evaluations one by one (a hand-off time-out shows as one 250 ms evaluation followed by
plain-schedule times); COCONS_DEBUG_ABORT=1 names the wait."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import cocons_amd as ca
from cocons_amd import workloads as wl

g = int(sys.argv[1]) if len(sys.argv) > 1 else 100
locs = wl.grid_locs(g)
X = wl.design_from_locs(locs)["std.covs"]
th = wl.theta_full()
z = wl.synthetic_z(g * g)
f = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
for i in range(8):
    t = time.perf_counter()
    v = f.neg2loglik_core(th)
    print("eval %d: %.2f ms  value %.6f" % (i, 1e3 * (time.perf_counter() - t), v[0] if isinstance(v, tuple) else v), flush=True)
