#!/usr/bin/env python3
"""Generate high-precision golden vectors for the dense hot path (mpmath, 40 digits).

The reference holds NO numeric golden vectors for cov_rns* / -2loglik
(tests/coco_test.R checks shapes, eigenvalues > 0 and non-NA only), and cannot be
built or run here (no R, Rcpp, Boost).  These vectors therefore come from an
independent exact-arithmetic evaluation of the *mathematical* model the
reference's code implements (src/cocons_full.cpp:40-594, formulas transcribed in
SURVEY.md §8a#6), not from the reference binary.

Outputs (small JSON, committed):
  besselk_grid.json      K_nu(x) on a (nu, x) grid incl. x->eps, x~2, x in [700,706)
  cov_nonstat_n20.json   full nonstationary Sigma (cov_rns), classic (cov_rns_classic)
                         and cross-covariance (cov_rns_pred), n=20, m=7, p=3
  neg2loglik_n20.json    sum(log diag chol), quadratic form and -2 loglik of that Sigma

Usage: python tests/golden/make_golden.py
"""
import json
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 40
HERE = os.path.dirname(os.path.abspath(__file__))


def fl(x):
    return float(x)


# --------------------------------------------------------------------------- #
def besselk_grid():
    nus = [0.25, 0.5, 0.5000001, 0.75, 0.999999, 1.0, 1.000001, 1.1180339887, 1.25, 1.5,
           1.73, 1.999, 2.0, 2.2360679775, 2.4999, 2.5, 3.0, 3.3]
    xs = [2.3e-16, 1e-12, 1e-8, 1e-4, 1e-2, 0.1, 0.5, 1.0, 1.5, 1.99, 1.9999999, 2.0,
          2.0000001, 2.01, 2.5, 3.0, 5.0, 8.0, 13.0, 25.0, 60.0, 150.0, 400.0, 700.0, 703.5, 705.99]
    rows = []
    for nu in nus:
        for x in xs:
            k = mp.besselk(mp.mpf(nu), mp.mpf(x))
            # also the scaled Matern correlation 2^(1-nu)/Gamma(nu) x^nu K_nu(x)
            mat = mp.power(2, 1 - mp.mpf(nu)) / mp.gamma(mp.mpf(nu)) * mp.power(mp.mpf(x), mp.mpf(nu)) * k
            rows.append({"nu": nu, "x": x, "K": fl(k), "K_log10": fl(mp.log10(k)), "matern": fl(mat)})
    return rows


# --------------------------------------------------------------------------- #
def dot(xrow, b):
    return mp.fsum(mp.mpf(float(a)) * mp.mpf(float(c)) if np.isfinite(c) else
                   (mp.mpf(float(a)) * mp.mpf(c) if a != 0 else mp.mpf(0))
                   for a, c in zip(xrow, b))


def loc_params(theta, X, smooth_limits, classic):
    p = X.shape[1]
    scale_je = list(theta["scale"])
    scale_je[0] = 0.0
    out = []
    for w in range(X.shape[0]):
        xr = X[w]
        t = {}
        t["tilt"] = mp.pi / (1 + mp.exp(-dot(xr, theta["tilt"])))
        t["rd"] = mp.exp(2 * dot(xr, scale_je))
        t["an"] = mp.exp(dot(xr, theta["aniso"]))
        t["dets"] = mp.exp(2 * dot(xr, scale_je) + dot(xr, theta["aniso"]))
        t["sigma"] = mp.exp(mp.mpf("0.5") * dot(xr, theta["std.dev"]))
        ng = theta["nugget"]
        if np.isneginf(ng[0]):
            t["nugget"] = mp.mpf(0)
        else:
            t["nugget"] = mp.exp(dot(xr, ng))
        if classic:
            t["nu"] = mp.exp(dot(xr, theta["smooth"]))
        else:
            lo, hi = mp.mpf(float(smooth_limits[0])), mp.mpf(float(smooth_limits[1]))
            t["nu"] = (hi - lo) / (1 + mp.exp(-dot(xr, theta["smooth"]))) + lo
        t["diag"] = mp.exp(dot(xr, theta["std.dev"])) + t["nugget"]
        out.append(t)
    return out


def pair_value(a, b, dx, dy, gr, classic):
    s11 = (a["rd"] + b["rd"]) / 2
    s22 = (a["rd"] * a["an"] ** 2 + b["rd"] * b["an"] ** 2) / 2
    s12 = (a["rd"] * a["an"] * mp.cos(a["tilt"]) + b["rd"] * b["an"] * mp.cos(b["tilt"])) / 2
    det = s11 * s22 - s12 * s12
    nu = (a["nu"] + b["nu"]) / 2 if classic else mp.sqrt(a["nu"]) * mp.sqrt(b["nu"])
    q = s22 * dx * dx + s11 * dy * dy - 2 * s12 * dx * dy
    u = mp.sqrt(8 * nu / (gr * det)) * mp.sqrt(q)
    if u == 0:
        return a["diag"]
    m = mp.power(2, 1 - nu) / mp.gamma(nu) * mp.power(u, nu) * mp.besselk(nu, u)
    amp = mp.sqrt(a["dets"] * mp.sin(a["tilt"]) * b["dets"] * mp.sin(b["tilt"]))
    return m * a["sigma"] * b["sigma"] * amp / mp.sqrt(det)


def mp_cov(theta, locs, X, smooth_limits, classic=False):
    n = X.shape[0]
    lp = loc_params(theta, X, smooth_limits, classic)
    gr = mp.exp(2 * mp.mpf(float(theta["scale"][0])))
    S = mp.zeros(n, n)
    for i in range(n):
        S[i, i] = lp[i]["diag"]
        for j in range(i + 1, n):
            dx = mp.mpf(float(locs[i, 0])) - mp.mpf(float(locs[j, 0]))
            dy = mp.mpf(float(locs[i, 1])) - mp.mpf(float(locs[j, 1]))
            S[i, j] = S[j, i] = pair_value(lp[i], lp[j], dx, dy, gr, classic)
    return S


def mp_cov_pred(theta, locs, locs_pred, X, Xp, smooth_limits):
    n, m = X.shape[0], Xp.shape[0]
    lo = loc_params(theta, X, smooth_limits, False)
    lpp = loc_params(theta, Xp, smooth_limits, False)
    gr = mp.exp(2 * mp.mpf(float(theta["scale"][0])))
    C = mp.zeros(m, n)
    for i in range(m):
        for j in range(n):
            if locs_pred[i, 0] == locs[j, 0] and locs_pred[i, 1] == locs[j, 1]:
                C[i, j] = lpp[i]["diag"]
                continue
            dx = mp.mpf(float(locs_pred[i, 0])) - mp.mpf(float(locs[j, 0]))
            dy = mp.mpf(float(locs_pred[i, 1])) - mp.mpf(float(locs[j, 1]))
            C[i, j] = pair_value(lpp[i], lo[j], dx, dy, gr, False)
    return C


def tolist(M):
    return [[fl(M[i, j]) for j in range(M.cols)] for i in range(M.rows)]


def main():
    with open(os.path.join(HERE, "besselk_grid.json"), "w") as f:
        json.dump(besselk_grid(), f)

    rng = np.random.default_rng(20251114)
    n, m, p = 20, 7, 3
    locs = rng.uniform(0, 1, size=(n, 2))
    locs_pred = rng.uniform(0, 1, size=(m, 2))
    locs_pred[2] = locs[5]                     # one exactly coincident point (cocons_full.cpp:410-414)
    cov = rng.normal(size=(n, 2))
    covp = rng.normal(size=(m, 2))
    covp[2] = cov[5]
    X = np.column_stack([np.ones(n), cov])
    Xp = np.column_stack([np.ones(m), covp])
    theta = {
        "mean": [0.3, -0.2, 0.1],
        "std.dev": [0.0, 0.3, -0.2],
        "scale": [float(np.log(0.3)), 0.2, 0.1],
        "aniso": [0.0, 0.25, -0.25],
        "tilt": [0.0, 0.3, 0.3],
        "smooth": [0.0, 0.5, -0.5],
        "nugget": [float(np.log(1e-2)), 0.1, 0.0],
    }
    theta_classic = dict(theta)
    theta_classic["smooth"] = [float(np.log(1.2)), 0.2, -0.1]
    sl = [0.5, 2.5]
    z = rng.normal(size=n)

    S = mp_cov(theta, locs, X, sl, classic=False)
    Sc = mp_cov(theta_classic, locs, X, sl, classic=True)
    C = mp_cov_pred(theta, locs, locs_pred, X, Xp, sl)
    theta_nonug = dict(theta)
    theta_nonug["nugget"] = [float("-inf"), 0.0, 0.0]
    S0 = mp_cov(theta_nonug, locs, X, sl, classic=False)

    def enc(t):
        return {k: [("-inf" if (isinstance(v, float) and np.isneginf(v)) else v) for v in vs] for k, vs in t.items()}

    with open(os.path.join(HERE, "cov_nonstat_n20.json"), "w") as f:
        json.dump({
            "locs": locs.tolist(), "locs_pred": locs_pred.tolist(), "X": X.tolist(), "X_pred": Xp.tolist(),
            "theta": enc(theta), "theta_classic": enc(theta_classic), "theta_nonugget": enc(theta_nonug),
            "smooth_limits": sl,
            "cov_rns": tolist(S), "cov_rns_classic": tolist(Sc), "cov_rns_pred": tolist(C),
            "cov_rns_nonugget": tolist(S0),
        }, f)

    # -2 loglik pieces from an exact Cholesky of the exact Sigma
    L = mp.cholesky(S)
    logdet_half = mp.fsum(mp.log(L[i, i]) for i in range(n))
    trend = [dot(X[i], theta["mean"]) for i in range(n)]
    r = mp.matrix([mp.mpf(float(z[i])) - trend[i] for i in range(n)])
    y = mp.lu_solve(L, r)          # L is lower triangular; LU of it is exact enough at 40 digits
    quad = mp.fsum(v * v for v in y)
    val = n * mp.log(2 * mp.pi) + 2 * logdet_half + quad
    with open(os.path.join(HERE, "neg2loglik_n20.json"), "w") as f:
        json.dump({"z": z.tolist(), "logdet_half": fl(logdet_half), "quad": fl(quad),
                   "neg2loglik_nopen": fl(val)}, f)
    print("golden vectors written")


if __name__ == "__main__":
    main()
