#!/usr/bin/env python3
"""Optimizer-in-the-loop harness (BASELINE config C4): reproduces the call pattern of
`optimParallel(method="L-BFGS-B")` as cocoOptim configures it (R/profile.R:11-18,
R/optim.R:237-259): central-difference gradient with ndeps = eps^(1/4) (forward = FALSE),
i.e. 1 + 2P objective evaluations per L-BFGS-B gradient request, lmm = 100, factr = 1e-8/eps.

  python tools/optim_loop.py [--g 64] [--evals 50]      -> prints evals/s on the GPU
"""
import argparse
import os
import sys
import time

import numpy as np
from scipy.optimize import minimize     # (imported HERE: inside lbfgsb_central the first call paid scipy.optimize's import --
#                                         a quarter of a second -- inside every timing of the loop: rounds 1-4's "C4 sequential
#                                         154-180 evals/s" against 433 for the same evaluations outside the loop was that import)

NDEPS = np.finfo(float).eps ** 0.25


def lbfgsb_central(fn, x0, lower, upper, max_evals=50, log=None, fn_batch=None):
    """fn_batch (optional): evaluates a list of points at once (the 2P gradient points, which
    optimParallel hands to its workers in parallel)."""
    count = {"n": 0}

    class Stop(Exception):
        pass

    best = {"f": np.inf, "x": np.array(x0, float)}

    def f(x):
        if count["n"] >= max_evals:
            raise Stop()
        count["n"] += 1
        v = fn(x)
        if v < best["f"]:
            best["f"], best["x"] = v, np.array(x, float)
        if log is not None:
            log.append(v)
        return v

    def fg(x):
        g = np.zeros_like(x)
        if fn_batch is not None:                      # 1 + 2P points in one pipelined batch
            if count["n"] + 1 + 2 * x.size > max_evals + 2 * x.size:
                raise Stop()
            pts = [x]
            for i in range(x.size):
                e = np.zeros_like(x)
                e[i] = NDEPS
                pts += [x + e, x - e]
            vals = fn_batch(pts)
            count["n"] += len(pts)
            if vals[0] < best["f"]:
                best["f"], best["x"] = vals[0], np.array(x, float)
            for i in range(x.size):
                g[i] = (vals[1 + 2 * i] - vals[2 + 2 * i]) / (2 * NDEPS)
            return vals[0], g
        f0 = f(x)
        for i in range(x.size):                       # the 2P points optimParallel farms out
            e = np.zeros_like(x)
            e[i] = NDEPS
            g[i] = (f(x + e) - f(x - e)) / (2 * NDEPS)
        return f0, g

    try:
        minimize(fg, x0, jac=True, method="L-BFGS-B", bounds=list(zip(lower, upper)),
                 options={"maxcor": 100, "ftol": 1e-8, "maxiter": 500})
    except Stop:
        pass
    return {"x": best["x"], "fun": best["f"], "nfev": count["n"]}


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    ap = argparse.ArgumentParser()
    ap.add_argument("--g", type=int, default=64)
    ap.add_argument("--evals", type=int, default=50)
    a = ap.parse_args()
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs = wl.grid_locs(a.g)
    n = a.g * a.g
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full()
    z = wl.synthetic_z(n)
    pp = wl.par_pos_full()
    t0 = wl.theta_vector_from_lists(th, pp) + 0.1
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)

    def fn(t):
        return ca.GetNeg2loglikelihood(t, pp, locs, X, wl.SMOOTH_LIMITS, z, n, (0, 0, 0), fit=fit)

    def fnb(ts):
        return ca.GetNeg2loglikelihood_batch(ts, pp, locs, X, wl.SMOOTH_LIMITS, z, n, (0, 0, 0), fit=fit)

    fn(t0)
    t = time.perf_counter()
    res = lbfgsb_central(fn, t0, t0 - 3, t0 + 3, max_evals=a.evals)
    dt = time.perf_counter() - t
    print("C4 sequential: n=%d P=%d evals=%d  %.2f evals/s (%.2f ms/eval)  f: %.6f -> %.6f" %
          (n, t0.size, res["nfev"], res["nfev"] / dt, 1e3 * dt / res["nfev"], fn(t0), res["fun"]))
    fnb([t0, t0])
    t = time.perf_counter()
    res = lbfgsb_central(fn, t0, t0 - 3, t0 + 3, max_evals=a.evals, fn_batch=fnb)
    dt = time.perf_counter() - t
    print("C4 batched gradient points: evals=%d  %.2f evals/s (%.2f ms/eval)  f -> %.6f" %
          (res["nfev"], res["nfev"] / dt, 1e3 * dt / res["nfev"], res["fun"]))


if __name__ == "__main__":
    main()
