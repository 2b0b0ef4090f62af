#!/usr/bin/env python3
"""Soak run on one GPU: many sequential evaluations at n = 10 000 with a parameter vector that keeps changing (as an optimiser's
does), interleaved with Profile / kriging calls on the same handle (which re-use and dirty the rows under the matrix), checking
every value against the first evaluation of the same parameters and that no engine hand-off ever timed out.
    python tools/soak.py [evaluations = 3000]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocons_amd as ca                     # noqa: E402
from cocons_amd import workloads as wl     # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
if os.environ.get("COCONS_SOAK_TRACE"):          # per-task stamps of the DAG launch: a time-out then lists the unfinished tasks
    from cocons_amd import _lib
    _lib.check(_lib.load().cocons_debug_tune(b"dag_trace", 1), "tune")
g = 100
locs = wl.grid_locs(g)
sc = wl.design_from_locs(locs)
X = sc["std.covs"]
z = wl.synthetic_z(g * g)
fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS, x_betas=X)
rng = np.random.default_rng(1)
base = wl.theta_full()
thetas = []
for i in range(8):
    t = {k: np.array(v, dtype=float) for k, v in base.items()}
    t["std.dev"][0] += 0.05 * rng.standard_normal()
    t["scale"][0] += 0.1 * rng.standard_normal()
    t["smooth"][0] += 0.1 * rng.standard_normal()
    thetas.append(t)
ref = [fit.neg2loglik_core(t)[0] for t in thetas]
lp = locs[:200] + 0.3 / (g - 1)
Xp = wl.design_from_locs(lp, sc["mean.vector"], sc["sd.vector"])["std.covs"]
pred_ref = fit.predict_core(thetas[0], lp, Xp)[0].copy()
t0 = time.perf_counter()
worst = 0.0
slow = []                                        # the slowest evaluations (ms, index): a paused queue shows up here
for i in range(N):
    k = int(rng.integers(0, 8))
    t1 = time.perf_counter()
    v = fit.neg2loglik_core(thetas[k])[0]
    slow.append(((time.perf_counter() - t1) * 1e3, i))
    if len(slow) > 64:
        slow = sorted(slow, reverse=True)[:5]
    worst = max(worst, abs(v - ref[k]) / abs(ref[k]))
    if i % 250 == 249:
        s = fit.predict_core(thetas[0], lp, Xp)[0]           # dirties the rows under the matrix, grows the border
        worst = max(worst, float(np.max(np.abs(s - pred_ref)) / np.max(np.abs(pred_ref))))
        print("%5d evaluations, %.1f evals/s so far, worst relative deviation %.2e, engine %s" %
              (i + 1, (i + 1) / (time.perf_counter() - t0), worst, fit.engine_state()), flush=True)
st = fit.engine_state()
print("slowest evaluations (ms, index): %s" % ", ".join("%.1f @ %d" % e for e in sorted(slow, reverse=True)[:5]))
print("done: %d evaluations in %.1f s; worst relative deviation from the first evaluation of the same parameters %.2e; engine %s"
      % (N, time.perf_counter() - t0, worst, st))
sys.exit(0 if (st["retries"] == 0 and worst < 1e-12) else 1)
