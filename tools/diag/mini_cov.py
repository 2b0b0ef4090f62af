import sys, numpy as np, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cocons_amd as ca
from cocons_amd import workloads as wl
rng = np.random.default_rng(6)
n = 257
locs = rng.uniform(0, 1, size=(n, 2))
X = wl.design_from_locs(locs)["std.covs"]
th = wl.theta_full(scale0=np.log(0.2))
which = sys.argv[1]
print("start", which, flush=True)
if which == "classic":
    thc = dict(th); thc["smooth"] = np.array([np.log(1.2), 0.2, -0.1])
    S = ca.cov_rns_classic(thc, locs, X)
elif which == "pred":
    lp = rng.uniform(0, 1, size=(190, 2)); Xp = wl.design_from_locs(lp)["std.covs"]
    S = ca.cov_rns_pred(th, locs, lp, X, Xp, wl.SMOOTH_LIMITS)
else:
    S = ca.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
print("done", which, np.isfinite(S).all(), flush=True)
