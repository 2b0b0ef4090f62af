#!/usr/bin/env python3
"""Golden vectors for the sparse/taper covariance entries (mpmath, 40 digits): an independent
exact-arithmetic evaluation of the model src/cocons_taper.cpp implements (isotropic nonstationary
Matern, local range r = exp(2 scale' x), sigma = exp(std.dev' x / 2), nu_ij = sqrt(nu_i nu_j)):

    C_ij = [2 sqrt(r_i r_j) / (r_i + r_j)] * sigma_i sigma_j * M_nu(sqrt(8 nu) |s_i - s_j| / sqrt((r_i + r_j)/2))
    C_ii = exp(std.dev' x_i) + exp(nugget' x_i)

The reference holds no numeric fixture for these functions either.  Output: taper_n24.json (CSR pattern,
1-based like spam, entries for: general nu, fixed nu = 1.5, and the prediction variant with one coincident
location).  Usage: python tests/golden/make_golden_taper.py
"""
import json
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 40
HERE = os.path.dirname(os.path.abspath(__file__))


def dot(xrow, b):
    return mp.fsum(mp.mpf(float(a)) * mp.mpf(float(c)) for a, c in zip(xrow, b))


def matern(nu, u):
    return mp.power(2, 1 - nu) / mp.gamma(nu) * mp.power(u, nu) * mp.besselk(nu, u)


def params(theta, X, sl, fixed_nu):
    out = []
    for row in X:
        r = mp.exp(2 * dot(row, theta["scale"]))
        sig = mp.exp(dot(row, theta["std.dev"]) / 2)
        if fixed_nu is None:
            nu = (mp.mpf(sl[1]) - mp.mpf(sl[0])) / (1 + mp.exp(-dot(row, theta["smooth"]))) + mp.mpf(sl[0])
        else:
            nu = mp.mpf(fixed_nu)
        diag = mp.exp(dot(row, theta["std.dev"])) + mp.exp(dot(row, theta["nugget"]))
        out.append((r, sig, nu, diag))
    return out


def entry(pa, pb, sa, sb):
    r1, s1, n1, _ = pa
    r2, s2, n2, _ = pb
    nu = mp.sqrt(n1 * n2)
    d = mp.sqrt((mp.mpf(float(sa[0])) - mp.mpf(float(sb[0]))) ** 2 + (mp.mpf(float(sa[1])) - mp.mpf(float(sb[1]))) ** 2)
    u = mp.sqrt(8 * nu) * d / mp.sqrt((r1 + r2) / 2)
    return 2 * mp.sqrt(r1 * r2) / (r1 + r2) * s1 * s2 * matern(nu, u)


def csr(rows_locs, cols_locs, delta):
    ci, rp = [], [1]
    for a in rows_locs:
        for j, b in enumerate(cols_locs):
            if np.hypot(a[0] - b[0], a[1] - b[1]) <= delta:
                ci.append(j + 1)
        rp.append(len(ci) + 1)
    return ci, rp


def main():
    rng = np.random.default_rng(20260301)
    n, m, p = 24, 9, 3
    locs = rng.uniform(0, 1, size=(n, 2))
    X = np.column_stack([np.ones(n), rng.standard_normal(n), rng.standard_normal(n)])
    lp = rng.uniform(0, 1, size=(m, 2))
    lp[4] = locs[7]                               # one coincident prediction location
    Xp = np.column_stack([np.ones(m), rng.standard_normal(m), rng.standard_normal(m)])
    theta = {"std.dev": [0.2, 0.3, -0.2], "scale": [float(np.log(0.3)), 0.15, 0.1], "aniso": [0.0, 0.0, 0.0],
             "tilt": [0.0, 0.0, 0.0], "smooth": [0.1, 0.5, -0.5], "nugget": [float(np.log(0.02)), 0.1, 0.0]}
    sl = [0.5, 2.5]
    ci, rp = csr(locs, locs, 0.45)
    P = params(theta, X, sl, None)
    gen = []
    for i in range(n):
        for w in range(rp[i] - 1, rp[i + 1] - 1):
            j = ci[w] - 1
            gen.append(float(P[i][3] if i == j else entry(P[i], P[j], locs[i], locs[j])))
    th15 = dict(theta)
    th15["smooth"] = [0.0, 0.0, 0.0]
    P15 = params(th15, X, [1.5, 1.5], 1.5)
    fix = []
    for i in range(n):
        for w in range(rp[i] - 1, rp[i + 1] - 1):
            j = ci[w] - 1
            fix.append(float(P15[i][3] if i == j else entry(P15[i], P15[j], locs[i], locs[j])))
    cip, rpp = csr(lp, locs, 0.5)
    Pp = params(theta, Xp, sl, None)
    pred = []
    for i in range(m):
        for w in range(rpp[i] - 1, rpp[i + 1] - 1):
            j = cip[w] - 1
            if lp[i][0] == locs[j][0] and lp[i][1] == locs[j][1]:
                pred.append(float(Pp[i][1] * Pp[i][1] + mp.exp(dot(Xp[i], theta["nugget"]))))
            else:
                pred.append(float(entry(Pp[i], P[j], lp[i], locs[j])))
    out = {"locs": locs.tolist(), "X": X.tolist(), "locs_pred": lp.tolist(), "X_pred": Xp.tolist(),
           "theta": theta, "smooth_limits": sl, "colindices": ci, "rowpointers": rp,
           "entries_general": gen, "entries_fixed_1p5": fix,
           "colindices_pred": cip, "rowpointers_pred": rpp, "entries_pred": pred}
    with open(os.path.join(HERE, "taper_n24.json"), "w") as fh:
        json.dump(out, fh)
    print("nnz", len(ci), "nnz_pred", len(cip))


if __name__ == "__main__":
    main()
