import sys, os, math
sys.path.insert(0, os.getcwd())
import numpy as np
import cocons_amd as ca
from cocons_amd import workloads as wl
g = 100
locs = wl.grid_locs(g); X = wl.design_from_locs(locs)["std.covs"]; th = wl.theta_full(); z = wl.synthetic_z(g*g)
def run(tag, fit, t):
    v = fit.neg2loglik_core(t)[0]
    print(tag, v, fit.engine_state(), flush=True)
f1 = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
run("r1 #1", f1, th); run("r1 #2", f1, th)
f2 = ca.CoconsFit(locs, X, np.column_stack([z, 3.0*z]), wl.SMOOTH_LIMITS)
run("r2 #1", f2, th); run("r2 #2", f2, th)
perm = np.random.default_rng(1).permutation(g*g)
f3 = ca.CoconsFit(locs[perm], X[perm], z[perm], wl.SMOOTH_LIMITS)
run("perm r1 #1", f3, th); run("perm r1 #2", f3, th)
th2 = {k: np.array(v, dtype=float) for k, v in th.items()}
th2["std.dev"][0] += 2 * math.log(1.7); th2["nugget"][0] += 2 * math.log(1.7)
run("r1 th2", f1, th2); run("r2 th2", f2, th2)
