"""CPU tests that pin the oracle: mpmath golden vectors, closed forms, cross-branch
identities, reference quirks.  No GPU needed."""
import json
import math
import os

import numpy as np
import pytest


def _dec(t):
    return {k: np.array([float(v) for v in vs]) for k, vs in t.items()}


def _relerr(a, b, floor=1e-290):
    a, b = np.asarray(a), np.asarray(b)
    mask = np.abs(b) > floor
    return float(np.max(np.abs(a[mask] - b[mask]) / np.abs(b[mask])))


def test_besselk_against_mpmath_grid(oracle, golden_dir):
    rows = json.load(open(os.path.join(golden_dir, "besselk_grid.json")))
    worst = 0.0
    for r in rows:
        if not (1e-300 < r["K"] < 1e300):
            continue
        k = oracle.besselk(r["nu"], r["x"])
        worst = max(worst, abs(k - r["K"]) / r["K"])
    assert worst < 2e-14, worst


def test_cov_golden_n20(oracle, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "cov_nonstat_n20.json")))
    locs, X = np.array(g["locs"]), np.array(g["X"])
    assert _relerr(oracle.cov_rns(_dec(g["theta"]), locs, X, g["smooth_limits"]), g["cov_rns"]) < 5e-14
    assert _relerr(oracle.cov_rns(_dec(g["theta_nonugget"]), locs, X, g["smooth_limits"]),
                   g["cov_rns_nonugget"]) < 5e-14
    assert _relerr(oracle.cov_rns_classic(_dec(g["theta_classic"]), locs, X), g["cov_rns_classic"]) < 5e-14
    C = oracle.cov_rns_pred(_dec(g["theta"]), locs, np.array(g["locs_pred"]), X, np.array(g["X_pred"]),
                            g["smooth_limits"])
    assert _relerr(C, g["cov_rns_pred"]) < 5e-14


def test_neg2loglik_golden_n20(oracle, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "cov_nonstat_n20.json")))
    n2 = json.load(open(os.path.join(golden_dir, "neg2loglik_n20.json")))
    th = _dec(g["theta"])
    locs, X = np.array(g["locs"]), np.array(g["X"])
    pp = {"mean": [True] * 3, "std.dev": [True] * 3, "scale": [True] * 3, "aniso": [True] * 3,
          "tilt": [True] * 3, "smooth": [True] * 3, "nugget": [True] * 3}
    sd, sc = th["std.dev"], th["scale"]
    raw = dict(th)
    raw["std.dev"], raw["scale"] = sd + sc, sd - sc
    tv = np.concatenate([raw[k] for k in pp])
    val = oracle.GetNeg2loglikelihood(tv, pp, locs, X, g["smooth_limits"], np.array(n2["z"]), 20, (0, 0, 0))
    assert abs(val - n2["neg2loglik_nopen"]) < 1e-12 * abs(n2["neg2loglik_nopen"])


@pytest.mark.parametrize("nu", [0.5, 1.5, 2.5])
def test_closed_form_stationary_matern(oracle, nu):
    """p=1, aniso=tilt=0, nugget=-Inf: Sigma_ij = e^sd * M_nu(sqrt(8 nu) h / e^scale0)."""
    rng = np.random.default_rng(5)
    n = 60
    locs = rng.uniform(0, 1, size=(n, 2))
    X = np.ones((n, 1))
    sd, sc = 0.3, math.log(0.2)
    th = {"std.dev": [sd], "scale": [sc], "aniso": [0.0], "tilt": [0.0], "smooth": [0.0], "nugget": [-np.inf]}
    S = oracle.cov_rns(th, locs, X, (nu, nu))
    h = np.sqrt(((locs[:, None, :] - locs[None, :, :]) ** 2).sum(-1))
    u = np.sqrt(8 * nu) * h / math.exp(sc)
    M = {0.5: np.exp(-u), 1.5: (1 + u) * np.exp(-u), 2.5: (1 + u + u * u / 3) * np.exp(-u)}[nu]
    want = math.exp(sd) * M
    assert _relerr(S, want) < 1e-13


def test_cross_branch_identity(oracle):
    """classic(log 1.5) == cov_rns(fixed 1.5) == cov_rns_pred(locs_pred = other points) pattern:
    Bessel branch vs closed form agree to ~1e-14."""
    rng = np.random.default_rng(6)
    n = 50
    locs = rng.uniform(0, 1, size=(n, 2))
    X = np.column_stack([np.ones(n), rng.standard_normal((n, 2))])
    th = {"std.dev": [0.1, 0.2, -0.1], "scale": [math.log(0.3), 0.1, 0.05], "aniso": [0.0, 0.2, -0.2],
          "tilt": [0.1, 0.3, 0.2], "smooth": [0.0, 0.0, 0.0], "nugget": [math.log(0.05), 0.0, 0.0]}
    a = oracle.cov_rns(th, locs, X, (1.5, 1.5))
    thc = dict(th)
    thc["smooth"] = [math.log(1.5), 0.0, 0.0]
    b = oracle.cov_rns_classic(thc, locs, X)
    assert _relerr(b, a) < 1e-13
    c = oracle.cov_rns_pred(th, locs[:30], locs[30:], X[:30], X[30:], (1.5, 1.5))
    # pred(i, j): ii = pred location (30+i), jj = obs j ; cov_rns used ii=j<jj=30+i: compare symmetric value
    assert _relerr(c, a[30:, :30]) < 1e-13


def test_quirks(oracle):
    rng = np.random.default_rng(7)
    n = 40
    locs = rng.uniform(0, 1, size=(n, 2))
    X = np.column_stack([np.ones(n), rng.standard_normal((n, 2))])
    th = {"std.dev": [0.1, 0.2, -0.1], "scale": [math.log(0.3), 0.1, 0.05], "aniso": [0.0, 0.2, -0.2],
          "tilt": [0.1, 0.3, 0.2], "smooth": [0.0, 0.0, 0.0], "nugget": [-np.inf, 0.0, 0.0]}
    # (i) fixed nu = 1.0: every off-diagonal equals the diagonal value of ii (min index)
    S = oracle.cov_rns(th, locs, X, (1.0, 1.0))
    assert S[5, 2] == S[2, 2] and S[2, 5] == S[2, 2]
    # nugget = -Inf encodes "no nugget": diag = Pexp(std.dev)
    assert S[3, 3] == 1 / math.exp(-(X[3] @ np.array(th["std.dev"])))
    # duplicate location: u <= eps -> diag of ii
    locs2 = locs.copy()
    locs2[9] = locs2[4]
    X2 = X.copy()
    X2[9] = X2[4]
    S2 = oracle.cov_rns(th, locs2, X2, (0.5, 2.5))
    assert S2[9, 4] == S2[4, 4]
    # u >= 706: asymptotic branch still finite / tiny
    th2 = dict(th)
    th2["scale"] = [math.log(0.0005), 0.0, 0.0]
    S3 = oracle.cov_rns(th2, locs, X, (0.5, 2.5))
    assert np.all(np.isfinite(S3)) and S3[0, 1] < 1e-200


def test_holes_c1_plumbing(oracle, golden_dir):
    """BASELINE config 1: 400 `holes` rows, stationary Matern nu=1.5, classic vs diff
    parameterisation agree and -2 loglik (LAPACK) matches the long-double truth."""
    d = np.loadtxt(os.path.join(golden_dir, "holes_train400.csv"), delimiter=",", skiprows=1)
    n = d.shape[0]
    locs, z = d[:, :2], d[:, 4]
    X = np.ones((n, 1))
    sc = math.log(0.2)
    th = {"mean": np.zeros(1), "std.dev": np.zeros(1), "scale": np.array([sc]), "aniso": np.zeros(1),
          "tilt": np.zeros(1), "smooth": np.zeros(1), "nugget": np.array([math.log(0.01)])}
    S = oracle.cov_rns(th, locs, X, (1.5, 1.5))
    thc = dict(th)
    thc["smooth"] = np.array([math.log(1.5)])
    Sc = oracle.cov_rns_classic(thc, locs, X)
    assert _relerr(Sc, S) < 1e-12
    pp = {"mean": 0.0, "std.dev": [True], "scale": [True], "aniso": 0.0, "tilt": 0.0, "smooth": 0.0,
          "nugget": [True]}
    tv = np.array([0.0 + sc, 0.0 - sc, math.log(0.01)])       # (sd', sc') = ((a+b)/2, (a-b)/2)
    val = oracle.GetNeg2loglikelihood(tv, pp, locs, X, (1.5, 1.5), z, n, (0, 0, 0))
    info, ld, quad, _ = oracle.chol_ld(S, z)
    assert info == 0
    truth = n * math.log(2 * math.pi) + 2 * ld + quad[0]
    assert abs(val - truth) < 1e-10 * abs(truth)


def test_penalty_and_theta_plumbing(oracle):
    pp = {"mean": [True, False, True], "std.dev": [True, True, False], "scale": [True, False, True],
          "aniso": 0.0, "tilt": 0.0, "smooth": 1.5, "nugget": -np.inf}
    theta = np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])
    tl = oracle.getModelLists(theta, pp, "diff")
    assert list(tl["mean"]) == [1.0, 0.0, 2.0]
    assert list(tl["std.dev"]) == [(3.0 + 5.0) / 2, 4.0, 0.0]     # only index 0 has both free
    assert list(tl["scale"]) == [(3.0 - 5.0) / 2, 0.0, 6.0]
    assert tl["smooth"][0] == 1.5 and np.isneginf(tl["nugget"][0])
    cl = oracle.getModelLists(theta, pp, "classic")
    assert list(cl["std.dev"]) == [3.0, 4.0, 0.0]
    # smoothed L1: |x| > 1e-4 -> |x| ; else softplus pair
    assert oracle.sumsmoothlone([0.5, -2.0], 2.0) == pytest.approx(5.0, rel=1e-15)
    small = oracle.sumsmoothlone([1e-6], 1.0)
    a = 1e6
    assert small == pytest.approx((math.log(1 + math.exp(-a * 1e-6)) + math.log(1 + math.exp(a * 1e-6))) / a)
    pen = oracle.getPen(10, (0.5, 0.25, 0.3), tl, (0.5, 2.5))
    zero = 2 * math.log(2.0) / 1e6              # smoothed |0|
    want = 0.3 * math.exp(tl["scale"][0]) * math.sqrt(2.0 / (1 + math.exp(-1.5)) + 0.5) + 0.25 * (2.0 + zero)
    want += 0.5 * (4.0 + 6.0 + 8 * zero)        # std.dev..smooth: entries 2..p, eight of them zero
    assert pen == pytest.approx(2 * 10 * want, rel=1e-14)
    X = np.column_stack([np.ones(5), np.arange(5.0), np.arange(5.0) ** 2])
    s = oracle.getScale(X)
    assert np.all(s["std.covs"][:, 0] == 1)
    assert abs(s["std.covs"][:, 1].mean()) < 1e-15 and s["std.covs"][:, 1].std(ddof=1) == pytest.approx(1.0)


def test_taper_golden_n24(oracle, golden_dir):
    """cov_rns_taper / cov_rns_taper_pred restatement against the 40-digit mpmath entries
    (tests/golden/make_golden_taper.py): general nu, fixed nu = 1.5, prediction variant incl. a
    coincident location."""
    g = json.load(open(os.path.join(golden_dir, "taper_n24.json")))
    th = {k: np.array(v) for k, v in g["theta"].items()}
    locs, X = np.array(g["locs"]), np.array(g["X"])
    a = oracle.cov_rns_taper(th, locs, X, g["colindices"], g["rowpointers"], g["smooth_limits"])
    assert _relerr(a, g["entries_general"]) < 5e-14
    th15 = dict(th)
    th15["smooth"] = np.zeros(3)
    b = oracle.cov_rns_taper(th15, locs, X, g["colindices"], g["rowpointers"], [1.5, 1.5])
    assert _relerr(b, g["entries_fixed_1p5"]) < 5e-14
    c = oracle.cov_rns_taper_pred(th, locs, np.array(g["locs_pred"]), X, np.array(g["X_pred"]),
                                  g["colindices_pred"], g["rowpointers_pred"], g["smooth_limits"])
    assert _relerr(c, g["entries_pred"]) < 5e-14
    # the dense and the taper functions agree where they model the same thing: no anisotropy, p = 1
    n = 12
    rng = np.random.default_rng(2)
    l2 = rng.uniform(0, 1, size=(n, 2))
    X1 = np.ones((n, 1))
    t1 = {"std.dev": np.array([0.3]), "scale": np.array([math.log(0.25)]), "aniso": np.zeros(1), "tilt": np.zeros(1),
          "smooth": np.zeros(1), "nugget": np.array([math.log(0.05)])}
    ci = np.tile(np.arange(1, n + 1), n)
    rp = np.arange(0, n + 1) * n + 1
    dense = oracle.cov_rns(t1, l2, X1, (1.5, 1.5))
    sparse = oracle.cov_rns_taper(t1, l2, X1, ci, rp, (1.5, 1.5)).reshape(n, n)
    assert _relerr(sparse, dense) < 1e-13


def test_taper_objective_independent_of_storage(oracle):
    """The oracle evaluates GetNeg2loglikelihoodTaper (R/neg2loglikelihood.R:20-53) with a DENSE Cholesky of the
    tapered matrix; the reference uses spam's sparse one.  log det and the quadratic form belong to the matrix:
    a sparse LU of the same CSR matrix (scipy, SuperLU -- spam is not available here) gives the same value."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    from cocons_amd import workloads as wl
    n = 400
    rng = np.random.default_rng(12)
    locs = rng.uniform(0, 1, size=(n, 2))
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full(scale0=math.log(0.2))
    delta = 0.2
    ci, rp, ent = [], [1], []
    for i in range(n):
        d = np.sqrt(np.sum((locs - locs[i]) ** 2, axis=1))
        idx = np.nonzero(d <= delta)[0]
        h = d[idx] / delta
        ci.extend((idx + 1).tolist())
        ent.extend(((1 - h) ** 4 * (4 * h + 1)).tolist())
        rp.append(len(ci) + 1)
    ref_taper = (np.array(ci, dtype=np.int32), np.array(rp, dtype=np.int32), np.array(ent))
    z = rng.standard_normal((n, 2))
    pp = wl.par_pos_full()
    tv = wl.theta_vector_from_lists(th, pp)
    lam = (0.0, 0.0, 0.0)
    got = oracle.GetNeg2loglikelihoodTaper(tv, pp, ref_taper, locs, X, wl.SMOOTH_LIMITS, z, n, lam)
    tl = oracle.getModelLists(tv, pp, "diff")
    vals = ref_taper[2] * oracle.cov_rns_taper(tl, locs, X, ref_taper[0], ref_taper[1], wl.SMOOTH_LIMITS)
    S = sp.csr_matrix((vals, ref_taper[0] - 1, ref_taper[1] - 1), shape=(n, n)).tocsc()
    assert abs(S - S.T).max() < 1e-15 * abs(S).max()
    lu = spl.splu(S)
    logdet = float(np.sum(np.log(np.abs(lu.U.diagonal())))) + float(np.sum(np.log(np.abs(lu.L.diagonal()))))
    want = 0.0
    for k in range(2):
        resid = z[:, k] - X @ tl["mean"]
        want += n * math.log(2 * math.pi) + logdet + float(resid @ lu.solve(resid))
    assert abs(got - want) < 1e-10 * abs(want)
    prof = oracle.GetNeg2loglikelihoodTaperProfile(tv, pp, ref_taper, locs, X, wl.SMOOTH_LIMITS, z, n, lam)
    assert np.isfinite(prof)
