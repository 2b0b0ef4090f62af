"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, the mpmath
golden vectors and closed forms.  Run on the GPU box with `pytest -m gpu`.

Tolerances (fp64 path; north star: -2 loglik within 1e-8 relative of the CPU path):
  ENTRY_RTOL  entrywise Sigma vs oracle, relative to the entry (closed-form and Bessel modes)
  N2LL_RTOL   -2 log-likelihood vs the CPU (LAPACK) path
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ENTRY_RTOL = 2e-12
N2LL_RTOL = 1e-8


def _problem(n, seed=0, scale0=np.log(0.2)):
    from cocons_amd import workloads as wl
    rng = np.random.default_rng(seed)
    locs = rng.uniform(0, 1, size=(n, 2))
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full(scale0=scale0)
    return locs, X, th, rng


def _relerr(a, b, floor=1e-290):
    a, b = np.asarray(a), np.asarray(b)
    mask = np.abs(b) > floor
    return float(np.max(np.abs(a[mask] - b[mask]) / np.abs(b[mask]))) if mask.any() else 0.0


@pytest.mark.parametrize("nu", [0.5, 1.5, 2.5])
def test_cov_rns_closed_form_modes(oracle, nu):
    import cocons_amd as ca
    locs, X, th, _ = _problem(301, seed=1)
    th["smooth"] = np.zeros(3)
    got = ca.cov_rns(th, locs, X, (nu, nu))
    want = oracle.cov_rns(th, locs, X, (nu, nu))
    assert got.shape == (301, 301)
    assert np.array_equal(got, got.T)
    assert _relerr(got, want) < ENTRY_RTOL


def test_cov_rns_general_bessel(oracle):
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs, X, th, _ = _problem(333, seed=2)
    got = ca.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
    want = oracle.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
    assert np.array_equal(got, got.T)
    assert _relerr(got, want) < ENTRY_RTOL


def test_cov_rns_small_range_large_u(oracle):
    """tiny range -> u up to and beyond 706 (asymptotic branch, cocons_full.cpp:301-305)."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs, X, th, _ = _problem(200, seed=3, scale0=np.log(0.004))
    got = ca.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
    want = oracle.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
    big = np.abs(want) > 1e-250
    assert _relerr(got[big], want[big]) < ENTRY_RTOL
    assert np.all(np.abs(got[~big] - want[~big]) < 1e-250)


def test_cov_rns_fixed_nu_quirk(oracle):
    """fixed smoothness not in {0.5,1.5,2.5}: the reference leaves the smooth vector at zero
    and every off-diagonal equals the ii diagonal value (SURVEY 8a#6 quirk i)."""
    import cocons_amd as ca
    locs, X, th, _ = _problem(130, seed=4)
    th["smooth"] = np.zeros(3)
    got = ca.cov_rns(th, locs, X, (1.0, 1.0))
    want = oracle.cov_rns(th, locs, X, (1.0, 1.0))
    assert _relerr(got, want) < ENTRY_RTOL
    i, j = 3, 77
    assert got[j, i] == got[i, i]


def test_cov_rns_no_nugget_and_duplicates(oracle):
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs, X, th, _ = _problem(150, seed=5)
    th["nugget"] = np.array([-np.inf, 0.0, 0.0])
    locs[10] = locs[3]
    X[10] = X[3]
    got = ca.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
    want = oracle.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
    assert _relerr(got, want) < ENTRY_RTOL
    assert got[10, 3] == got[3, 3]          # u <= eps -> diagonal value of ii


def test_cov_rns_classic_and_pred(oracle):
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs, X, th, rng = _problem(257, seed=6)
    thc = dict(th)
    thc["smooth"] = np.array([np.log(1.2), 0.2, -0.1])
    got = ca.cov_rns_classic(thc, locs, X)
    want = oracle.cov_rns_classic(thc, locs, X)
    assert _relerr(got, want) < ENTRY_RTOL
    m = 190
    lp = rng.uniform(0, 1, size=(m, 2))
    lp[5] = locs[17]
    Xp = wl.design_from_locs(lp)["std.covs"]
    gotp = ca.cov_rns_pred(th, locs, lp, X, Xp, wl.SMOOTH_LIMITS)
    wantp = oracle.cov_rns_pred(th, locs, lp, X, Xp, wl.SMOOTH_LIMITS)
    assert gotp.shape == (m, 257)
    assert _relerr(gotp, wantp) < ENTRY_RTOL


def test_golden_mpmath_n20(golden_dir):
    import cocons_amd as ca
    g = json.load(open(os.path.join(golden_dir, "cov_nonstat_n20.json")))

    def dec(t):
        return {k: np.array([float(v) for v in vs]) for k, vs in t.items()}

    locs, X = np.array(g["locs"]), np.array(g["X"])
    S = ca.cov_rns(dec(g["theta"]), locs, X, g["smooth_limits"])
    assert _relerr(S, np.array(g["cov_rns"])) < 1e-13
    Sc = ca.cov_rns_classic(dec(g["theta_classic"]), locs, X)
    assert _relerr(Sc, np.array(g["cov_rns_classic"])) < 1e-13
    C = ca.cov_rns_pred(dec(g["theta"]), locs, np.array(g["locs_pred"]), X, np.array(g["X_pred"]),
                        g["smooth_limits"])
    assert _relerr(C, np.array(g["cov_rns_pred"])) < 1e-13
    n2 = json.load(open(os.path.join(golden_dir, "neg2loglik_n20.json")))
    th = dec(g["theta"])
    fit = ca.CoconsFit(locs, X, np.array(n2["z"]), g["smooth_limits"])
    val, parts = fit.neg2loglik_core(th)
    assert abs(parts[0] - n2["logdet_half"]) < 1e-12 * max(1, abs(n2["logdet_half"]))
    assert abs(parts[1] - n2["quad"]) < 1e-11 * abs(n2["quad"])
    assert abs(val - n2["neg2loglik_nopen"]) < 1e-12 * abs(n2["neg2loglik_nopen"])


@pytest.mark.parametrize("n", [100, 128, 300, 513])
def test_chol_solve_vs_long_double(oracle, n):
    from cocons_amd import _lib
    import ctypes
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n))
    A = np.asfortranarray(B @ B.T + n * np.eye(n))
    rhs = np.asfortranarray(rng.standard_normal((n, 3)))
    L = np.zeros((n, n), order="F")
    Y = np.zeros((n, 3), order="F")
    ld = ctypes.c_double()
    lib = _lib.load()
    dp = ctypes.POINTER(ctypes.c_double)
    rc = lib.cocons_chol_solve(n, A.ctypes.data_as(dp), 3, rhs.ctypes.data_as(dp), L.ctypes.data_as(dp),
                               Y.ctypes.data_as(dp), ctypes.byref(ld))
    assert rc == 0, _lib.last_error()
    info, ld_true, quad, Ytrue = oracle.chol_ld(A, rhs)
    assert info == 0
    assert abs(ld.value - ld_true) < 1e-12 * abs(ld_true)
    assert np.max(np.abs(Y - Ytrue)) < 1e-11 * np.max(np.abs(Ytrue))
    assert np.max(np.abs(L @ L.T - A)) < 1e-12 * np.max(np.abs(A))


def test_chol_not_positive_definite():
    from cocons_amd import _lib
    import ctypes
    n = 200
    A = np.asfortranarray(np.eye(n))
    A[150, 150] = -1.0
    lib = _lib.load()
    dp = ctypes.POINTER(ctypes.c_double)
    ld = ctypes.c_double()
    rc = lib.cocons_chol_solve(n, A.ctypes.data_as(dp), 0, None, None, None, ctypes.byref(ld))
    assert rc == 151


@pytest.mark.parametrize("aspect,value", [("scale", 300.0), ("aniso", 400.0)])
def test_degenerate_parameters_poison_sigma_like_the_reference(oracle, aspect, value):
    """Overflowing link functions (exp(2 scale' x) = inf at some locations, 0 at others) make inf - inf in the averaged kernel
    matrix: the reference's sqrt / Bessel chain then returns NaN for those pairs (src/cocons_full.cpp:286-305; NaN fails both
    branch tests), the Cholesky fails and the objective's tryCatch contract fires (R/neg2loglikelihood.R:200-206).  The HIP path
    must produce NaN for exactly the pairs the CPU restatement does -- until round 4 its square root returned 0 for NaN, which
    sent such pairs down the `u <= epsilon` branch with the diagonal value -- and agree where the entry is finite."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs, X, th, rng = _problem(150, seed=0)
    th = {k: np.array(v, dtype=float) for k, v in th.items()}
    th[aspect][1] = value
    S = ca.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
    So = oracle.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
    assert np.isnan(So).mean() > 0.2                       # (the case is what it claims to be)
    assert np.array_equal(np.isnan(S), np.isnan(So))
    if aspect == "scale":
        # (where the entry is finite the two agree as usual; with exp(400 x) in the anisotropy the finite entries are products
        # of numbers near both ends of the exponent range, which the device's merged square roots / one reciprocal of det --
        # written for operands of ordinary magnitude, matern_device.hpp -- round to 0 or inf differently: not compared)
        fin = np.isfinite(So) & (np.abs(So) > 1e-280)
        assert np.max(np.abs(S[fin] - So[fin]) / np.abs(So[fin])) < ENTRY_RTOL
    pp = wl.par_pos_full()
    tv = wl.theta_vector_from_lists(th, pp)
    z = rng.standard_normal(150)
    assert oracle.GetNeg2loglikelihood(tv, pp, locs, X, wl.SMOOTH_LIMITS, z, 150, (0, 0, 0)) == 1e6
    assert ca.GetNeg2loglikelihood(tv, pp, locs, X, wl.SMOOTH_LIMITS, z, 150, (0, 0, 0)) == 1e6
    with pytest.raises(RuntimeError, match="Cholesky error"):
        ca.GetNeg2loglikelihood(tv, pp, locs, X, wl.SMOOTH_LIMITS, z, 150, (0, 0, 0), safe=False)


@pytest.mark.parametrize("n", [300, 700])
def test_neg2loglik_vs_cpu(oracle, n):
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs, X, th, rng = _problem(n, seed=n)
    z = rng.standard_normal((n, 2))
    pp = wl.par_pos_full()
    tv = wl.theta_vector_from_lists(th, pp)
    lam = (0.1, 0.2, 0.3)
    got = ca.GetNeg2loglikelihood(tv, pp, locs, X, wl.SMOOTH_LIMITS, z, n, lam)
    want = oracle.GetNeg2loglikelihood(tv, pp, locs, X, wl.SMOOTH_LIMITS, z, n, lam)
    assert abs(got - want) <= N2LL_RTOL * abs(want)


def test_neg2loglik_safe_sentinel():
    """Cholesky failure -> exactly 1e6 when safe, error otherwise (R/neg2loglikelihood.R:200-206)."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs, X, th, rng = _problem(140, seed=9)
    th["smooth"] = np.zeros(3)            # fixed nu = 1.0 quirk -> singular Sigma
    pp = wl.par_pos_full()
    pp["smooth"] = 0.0
    tv = wl.theta_vector_from_lists(th, pp)
    z = rng.standard_normal(140)
    assert ca.GetNeg2loglikelihood(tv, pp, locs, X, (1.0, 1.0), z, 140, (0, 0, 0)) == 1e6
    with pytest.raises(RuntimeError, match="Cholesky error"):
        ca.GetNeg2loglikelihood(tv, pp, locs, X, (1.0, 1.0), z, 140, (0, 0, 0), safe=False)


def test_profile_and_reml_vs_cpu(oracle):
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    n = 260
    locs, X, th, rng = _problem(n, seed=11)
    z = rng.standard_normal((n, 2)) + 0.5
    pp = wl.par_pos_full()
    tv = wl.theta_vector_from_lists(th, pp)
    lam = (0.1, 0.0, 0.3)
    got = ca.GetNeg2loglikelihoodProfile(tv, pp, locs, X, wl.SMOOTH_LIMITS, z, n, X, lam)
    want = oracle.GetNeg2loglikelihoodProfile(tv, pp, locs, X, wl.SMOOTH_LIMITS, z, n, X, lam)
    assert abs(got - want) <= N2LL_RTOL * abs(want)
    got = ca.GetNeg2loglikelihoodREML(tv, pp, locs, X, X, wl.SMOOTH_LIMITS, z, n, lam)
    want = oracle.GetNeg2loglikelihoodREML(tv, pp, locs, X, X, wl.SMOOTH_LIMITS, z, n, lam)
    assert abs(got - want) <= N2LL_RTOL * abs(want)


def test_profile_betas_vs_numpy(oracle):
    """beta recovery after a pml fit (R/optim.R:329-341) against the literal formula."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    n = 310
    locs, X, th, rng = _problem(n, seed=13)
    z = rng.standard_normal((n, 3)) + (X @ np.array([0.5, -0.3, 0.2]))[:, None]
    got = ca.getBetas_profile(th, locs, X, wl.SMOOTH_LIMITS, z, X)
    S = oracle.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
    V = np.linalg.solve(S, X)
    W = X.T @ V
    want = np.linalg.solve(W, V.T) @ z.sum(axis=1) / z.shape[1]
    assert np.max(np.abs(got - want)) < 1e-9 * np.max(np.abs(want))


def test_predict_vs_cpu(oracle):
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    n, m = 280, 150
    locs, X, th, rng = _problem(n, seed=12)
    th["mean"] = np.array([0.2, -0.1, 0.05])
    lp = rng.uniform(0, 1, size=(m, 2))
    lp[7] = locs[30]
    sc = wl.design_from_locs(locs)
    Xp = wl.design_from_locs(lp, sc["mean.vector"], sc["sd.vector"])["std.covs"]
    z = rng.standard_normal(n)
    got = ca.cocoPredict_dense(th, locs, lp, X, Xp, wl.SMOOTH_LIMITS, z)
    want = oracle.cocoPredict_dense(th, locs, lp, X, Xp, wl.SMOOTH_LIMITS, z)
    assert np.allclose(got["systematic"], want["systematic"], rtol=1e-13, atol=0)
    assert np.max(np.abs(got["stochastic"] - want["stochastic"])) < 1e-9 * np.max(np.abs(want["stochastic"]))
    # compare predictive VARIANCES: at the coincident point the variance is rounding noise
    # around 0 and the reference's sqrt(abs(.)) (R/predict.R:175-183) amplifies that noise
    vg, vw = got["sd.pred"] ** 2, want["sd.pred"] ** 2
    assert np.max(np.abs(vg - vw)) < 1e-11 * np.max(vw)
    assert got["sd.pred"][7] < 1e-6 and want["sd.pred"][7] < 1e-6


def test_batch_matches_single_calls(oracle):
    """cocons_neg2loglik_batch: pipelined independent evaluations give the values of the
    single calls; a theta whose Sigma is not positive definite yields 1e6 (safe) in place."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    n = 520
    locs, X, th, rng = _problem(n, seed=21)
    z = rng.standard_normal(n)
    pp = wl.par_pos_full()
    t0 = wl.theta_vector_from_lists(th, pp)
    thetas = [t0 + 0.05 * rng.standard_normal(t0.size) for _ in range(7)]
    bad = t0.copy()
    bad[-1] = -800.0                      # nugget -> 0 ...
    bad[0], bad[3] = 30.0, -30.0          # ... std.dev' = 0, scale' = 30: range e^30, Sigma = all-ones (rank 1)
    thetas.insert(3, bad)
    lam = (0.1, 0.1, 0.1)
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
    got = ca.GetNeg2loglikelihood_batch(thetas, pp, locs, X, wl.SMOOTH_LIMITS, z, n, lam, fit=fit)
    one = np.array([ca.GetNeg2loglikelihood(t, pp, locs, X, wl.SMOOTH_LIMITS, z, n, lam, fit=fit) for t in thetas])
    # the batch slots run the plain schedule, a lone evaluation the engine schedule (the diagonal block
    # is then updated in a different summation order): equal to rounding, not bit for bit
    assert np.allclose(got, one, rtol=1e-12, atol=0)
    assert got[3] == 1e6 and one[3] == 1e6
    want0 = oracle.GetNeg2loglikelihood(thetas[0], pp, locs, X, wl.SMOOTH_LIMITS, z, n, lam)
    assert abs(got[0] - want0) <= N2LL_RTOL * abs(want0)
    with pytest.raises(RuntimeError, match="Cholesky error"):
        ca.GetNeg2loglikelihood_batch(thetas, pp, locs, X, wl.SMOOTH_LIMITS, z, n, lam, safe=False, fit=fit)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 129, 200])
def test_ragged_sizes_all_entry_points(oracle, n):
    """sizes around the 64-wide pair tiles and the 128-wide factorisation tiles, p = 1 and p = 5,
    several realizations."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    rng = np.random.default_rng(1000 + n)
    locs = rng.uniform(0, 1, size=(n, 2))
    for p in (1, 5):
        X = np.column_stack([np.ones(n)] + [rng.standard_normal(n) * 0.5 for _ in range(p - 1)])
        th = {"mean": rng.standard_normal(p) * 0.1,
              "std.dev": np.r_[0.1, rng.standard_normal(p - 1) * 0.1],
              "scale": np.r_[np.log(0.3), rng.standard_normal(p - 1) * 0.1],
              "aniso": np.r_[0.0, rng.standard_normal(p - 1) * 0.1],
              "tilt": np.r_[0.1, rng.standard_normal(p - 1) * 0.1],
              "smooth": np.r_[0.2, rng.standard_normal(p - 1) * 0.2],
              "nugget": np.r_[np.log(0.05), np.zeros(p - 1)]}
        got = ca.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
        want = oracle.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
        assert got.shape == (n, n) and _relerr(got, want) < ENTRY_RTOL
        assert _relerr(ca.cov_rns_classic(th, locs, X), oracle.cov_rns_classic(th, locs, X)) < ENTRY_RTOL
        m = max(1, n // 3)
        lp = rng.uniform(0, 1, size=(m, 2))
        Xp = np.column_stack([np.ones(m)] + [rng.standard_normal(m) * 0.5 for _ in range(p - 1)])
        assert _relerr(ca.cov_rns_pred(th, locs, lp, X, Xp, wl.SMOOTH_LIMITS),
                       oracle.cov_rns_pred(th, locs, lp, X, Xp, wl.SMOOTH_LIMITS)) < ENTRY_RTOL
        z = rng.standard_normal((n, 3))
        fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
        val, parts = fit.neg2loglik_core(th)
        info, ld, quad, _ = oracle.chol_ld(want, z - (X @ th["mean"])[:, None])
        assert info == 0
        truth = sum(n * np.log(2 * np.pi) + 2 * ld + q for q in quad)
        assert abs(val - truth) <= 1e-9 * abs(truth)
        assert np.allclose(parts[1:], quad, rtol=1e-9)
        st, qf = fit.predict_core(th, lp, Xp, z_col=1)
        Cw = oracle.cov_rns_pred(th, locs, lp, X, Xp, wl.SMOOTH_LIMITS)
        sol = np.linalg.solve(want, Cw.T)
        assert np.allclose(st, (z[:, 1] - X @ th["mean"]) @ sol, rtol=1e-8, atol=1e-10)
        assert np.allclose(qf, np.sum(Cw * sol.T, axis=1), rtol=1e-8, atol=1e-12)


def test_widest_design_matrix(oracle):
    """p = COCONS_P_MAX = 32 covariates in every aspect (the widest design the ABI takes): entries, the prediction
    cross-covariance, the objective and the taper entries against the CPU restatement; p = 33 is refused."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    n, p = 300, 32
    rng = np.random.default_rng(3232)
    locs = rng.uniform(0, 1, size=(n, 2))
    X = np.column_stack([np.ones(n)] + [rng.standard_normal(n) * 0.3 for _ in range(p - 1)])
    sm = 0.03
    th = {"mean": rng.standard_normal(p) * 0.1,
          "std.dev": np.r_[0.1, rng.standard_normal(p - 1) * sm],
          "scale": np.r_[np.log(0.3), rng.standard_normal(p - 1) * sm],
          "aniso": np.r_[0.0, rng.standard_normal(p - 1) * sm],
          "tilt": np.r_[0.1, rng.standard_normal(p - 1) * sm],
          "smooth": np.r_[0.2, rng.standard_normal(p - 1) * sm],
          "nugget": np.r_[np.log(0.05), rng.standard_normal(p - 1) * sm]}
    want = oracle.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
    assert _relerr(ca.cov_rns(th, locs, X, wl.SMOOTH_LIMITS), want) < ENTRY_RTOL
    m = 40
    lp = rng.uniform(0, 1, size=(m, 2))
    Xp = np.column_stack([np.ones(m)] + [rng.standard_normal(m) * 0.3 for _ in range(p - 1)])
    assert _relerr(ca.cov_rns_pred(th, locs, lp, X, Xp, wl.SMOOTH_LIMITS),
                   oracle.cov_rns_pred(th, locs, lp, X, Xp, wl.SMOOTH_LIMITS)) < ENTRY_RTOL
    z = rng.standard_normal(n)
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
    val, parts = fit.neg2loglik_core(th)
    info, ld, quad, _ = oracle.chol_ld(want, (z - X @ th["mean"])[:, None])
    assert info == 0
    truth = n * np.log(2 * np.pi) + 2 * ld + quad[0]
    assert abs(val - truth) <= 1e-9 * abs(truth)
    ci, rp = _csr_within(locs, locs, 0.2)
    assert _relerr(ca.cov_rns_taper(th, locs, X, ci, rp, wl.SMOOTH_LIMITS),
                   oracle.cov_rns_taper(th, locs, X, ci, rp, wl.SMOOTH_LIMITS)) < ENTRY_RTOL
    X33 = np.column_stack([X, rng.standard_normal(n)])
    th33 = {k: np.r_[v, 0.0] for k, v in th.items()}
    with pytest.raises(ValueError, match="up to 32"):
        ca.cov_rns(th33, locs, X33, wl.SMOOTH_LIMITS)


def test_handle_refuses_use_after_fork():
    """cocoOptim forks its workers (R/optim.R:117-121); a HIP context does not survive fork, so a
    handle created in the parent must be refused in the child (before any HIP call)."""
    import multiprocessing as mp
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs, X, th, rng = _problem(50, seed=3)
    fit = ca.CoconsFit(locs, X, rng.standard_normal(50), wl.SMOOTH_LIMITS)
    fit.neg2loglik_core(th)

    def child(q):
        import os
        try:
            fit.neg2loglik_core(th)
            q.put("ran")
        except Exception as e:            # noqa: BLE001
            q.put(str(e))
        q.close()
        q.join_thread()
        os._exit(0)                       # no atexit / HIP runtime teardown in the forked child

    ctx = mp.get_context("fork")
    q = ctx.Queue()
    pr = ctx.Process(target=child, args=(q,))
    pr.start()
    msg = q.get(timeout=60)
    pr.join(timeout=60)
    assert "another process" in msg


@pytest.mark.parametrize("type_", ["classic", "diff"])
def test_cocoSim_marginal_vs_cpu(oracle, type_):
    """cocoSim, marginal branch: same N(0,1) draws -> same fields."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    n = 333
    locs, X, th, rng = _problem(n, seed=31)
    th["mean"] = np.array([0.3, -0.2, 0.1])
    if type_ == "classic":
        th["smooth"] = np.array([np.log(1.2), 0.2, -0.1])
    E = rng.standard_normal((n, 11))
    got = ca.cocoSim_dense(th, locs, X, wl.SMOOTH_LIMITS, E, type=type_)
    want = oracle.cocoSim_dense(th, locs, X, wl.SMOOTH_LIMITS, E, type=type_)
    assert got.shape == (n, 11)
    assert np.max(np.abs(got - want)) < 1e-10 * np.max(np.abs(want))


def test_cocoSim_conditional_vs_cpu(oracle):
    """cocoSim, conditional branch (R/sim.R:69-127): one joint Cholesky against the literal
    solve + Schur complement + chol + cocoPredict(type='mean') of the CPU restatement."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    n, m = 300, 170
    locs, X, th, rng = _problem(n, seed=41)
    th["mean"] = np.array([0.3, -0.2, 0.1])
    lp = rng.uniform(0, 1, size=(m, 2))
    sc = wl.design_from_locs(locs)
    Xp = wl.design_from_locs(lp, sc["mean.vector"], sc["sd.vector"])["std.covs"]
    newdataset = np.column_stack([lp, rng.standard_normal((m, 2))])     # x, y, covariates...
    z = rng.standard_normal(n)
    E = rng.standard_normal((m, 5))
    got = ca.cocoSim_cond_dense(th, locs, lp, newdataset, X, Xp, wl.SMOOTH_LIMITS, z, E)
    want = oracle.cocoSim_cond_dense(th, locs, lp, newdataset, X, Xp, wl.SMOOTH_LIMITS, z, E)
    assert got.shape == (m, 5)
    assert np.max(np.abs(got - want)) < 1e-8 * np.max(np.abs(want))


def test_evaluation_after_predict_uses_small_border(oracle):
    """a predict call grows the buffer under the matrix; later evaluations on the same handle must
    still give the same value (and only carry their own right-hand-side rows)."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    n, m = 260, 700
    locs, X, th, rng = _problem(n, seed=51)
    z = rng.standard_normal(n)
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
    v0, p0 = fit.neg2loglik_core(th)
    lp = rng.uniform(0, 1, size=(m, 2))
    Xp = wl.design_from_locs(lp)["std.covs"]
    fit.predict_core(th, lp, Xp)
    v1, p1 = fit.neg2loglik_core(th)
    assert v1 == v0 and np.array_equal(p0, p1)


def test_fuzz_random_parameters(oracle):
    """30 random parameter sets (ranges 0.03..0.6, nuggets 1e-4..0.1, smoothness limits varied,
    strong covariate effects) at n = 400: every GPU value within the north-star tolerance of the CPU
    path, or both sides agree that the Cholesky fails."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    rng = np.random.default_rng(2025)
    n = 400
    locs = rng.uniform(0, 1, size=(n, 2))
    X = wl.design_from_locs(locs)["std.covs"]
    z = rng.standard_normal(n)
    pp = wl.par_pos_full()
    worst = 0.0
    nfail = 0
    for it in range(30):
        lim = [(0.5, 2.5), (0.3, 1.2), (1.0, 3.0)][it % 3]
        th = wl.theta_full(scale0=np.log(rng.uniform(0.03, 0.6)))
        for k in ("std.dev", "scale", "aniso", "tilt", "smooth"):
            th[k] = th[k] + np.r_[0.0, rng.normal(0, 0.4, size=2)]
        th["nugget"] = np.array([np.log(10 ** rng.uniform(-4, -1)), 0.0, 0.0])
        tv = wl.theta_vector_from_lists(th, pp)
        fit = ca.CoconsFit(locs, X, z, lim)
        got = ca.GetNeg2loglikelihood(tv, pp, locs, X, lim, z, n, (0.1, 0.1, 0.1), fit=fit)
        want = oracle.GetNeg2loglikelihood(tv, pp, locs, X, lim, z, n, (0.1, 0.1, 0.1))
        if want == 1e6 or got == 1e6:
            assert got == want, (it, got, want)
            nfail += 1
            continue
        worst = max(worst, abs(got - want) / abs(want))
    assert worst <= N2LL_RTOL, worst
    assert nfail < 10


def test_spatial_sort_is_transparent(oracle, monkeypatch):
    """The handle stores the observations in Morton order (faster Bessel kernels on scattered data).
    Scalars, kriging outputs and conditional fields must not depend on it; the marginal simulation
    must come back in the caller's order for the caller's draws."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    n, m = 500, 60
    locs, X, th, rng = _problem(n, seed=61)
    th["mean"] = np.array([0.2, 0.1, -0.1])
    z = rng.standard_normal(n)
    lp = rng.uniform(0, 1, size=(m, 2))
    Xp = wl.design_from_locs(lp)["std.covs"]
    E = rng.standard_normal((n, 3))
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("COCONS_SPATIAL_SORT", flag)
        fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
        res[flag] = (fit.neg2loglik_core(th)[0], fit.predict_core(th, lp, Xp), fit.sim_core(th, E))
        fit.close()
    assert abs(res["0"][0] - res["1"][0]) < 1e-11 * abs(res["0"][0])
    assert np.allclose(res["0"][1][0], res["1"][1][0], rtol=1e-9, atol=1e-12)
    assert np.allclose(res["0"][1][1], res["1"][1][1], rtol=1e-9, atol=1e-12)
    assert np.array_equal(res["0"][2], res["1"][2])          # same (unsorted) path either way
    want = oracle.cocoSim_dense(th, locs, X, wl.SMOOTH_LIMITS, E, type="diff")
    assert np.max(np.abs(res["1"][2] - want)) < 1e-10 * np.max(np.abs(want))


def test_device_matern_against_mpmath_grid(golden_dir):
    """The DEVICE Bessel/Matern routine (Temme series / CF2 in A-B product form / rgamma table /
    exp2 -- a different algorithmic form from the oracle's) evaluated pointwise on the 468-point
    mpmath grid: nu 0.25..3.3, x 2.3e-16..705.99, including the x = 2 switch."""
    import json
    from cocons_amd import _lib
    rows = json.load(open(os.path.join(golden_dir, "besselk_grid.json")))
    nu = np.array([r["nu"] for r in rows])
    x = np.array([r["x"] for r in rows])
    want = np.array([r["matern"] for r in rows])
    out = np.empty_like(x)
    L = _lib.load()
    _lib.check(L.cocons_debug_matern(x.size, nu.ctypes.data_as(_lib.c_dp), x.ctypes.data_as(_lib.c_dp),
                                     out.ctypes.data_as(_lib.c_dp)), "cocons_debug_matern")
    ok = want > 1e-290          # below that the product underflows gradually in both
    rel = np.abs(out[ok] - want[ok]) / want[ok]
    assert rel.max() < 1e-13, (rel.max(), nu[ok][rel.argmax()], x[ok][rel.argmax()])
    assert np.all(np.abs(out[~ok] - want[~ok]) < 1e-290)


def _csr_within(rows, cols, delta):
    ci, rp = [], [1]
    for a in rows:
        d = np.hypot(cols[:, 0] - a[0], cols[:, 1] - a[1])
        ci.extend((np.nonzero(d <= delta)[0] + 1).tolist())
        rp.append(len(ci) + 1)
    return np.array(ci, dtype=np.int32), np.array(rp, dtype=np.int32)


def test_taper_entries_vs_oracle_and_golden(oracle, golden_dir):
    """cov_rns_taper / cov_rns_taper_pred (assembly slice): mpmath golden entries at
    n = 24 and the CPU restatement entry by entry on a 3000-point pattern (~90 neighbours per row),
    every smoothness branch, the fixed-nu quirk, an empty row and a coincident prediction location."""
    import cocons_amd as ca
    g = json.load(open(os.path.join(golden_dir, "taper_n24.json")))
    th = {k: np.array(v) for k, v in g["theta"].items()}
    locs, X = np.array(g["locs"]), np.array(g["X"])
    a = ca.cov_rns_taper(th, locs, X, g["colindices"], g["rowpointers"], g["smooth_limits"])
    assert _relerr(a, g["entries_general"]) < 1e-13
    c = ca.cov_rns_taper_pred(th, locs, np.array(g["locs_pred"]), X, np.array(g["X_pred"]),
                              g["colindices_pred"], g["rowpointers_pred"], g["smooth_limits"])
    assert _relerr(c, g["entries_pred"]) < 1e-13
    n = 3000
    locs, X, th, rng = _problem(n, seed=31)
    ci, rp = _csr_within(locs, locs, 0.1)
    assert ci.size > 50 * n
    got = ca.cov_rns_taper(th, locs, X, ci, rp, wl_limits())
    want = oracle.cov_rns_taper(th, locs, X, ci, rp, wl_limits())
    assert _relerr(got, want) < ENTRY_RTOL
    for nu in (0.5, 1.5, 2.5, 1.0):                      # closed forms and the degenerate fixed-nu quirk
        t2 = {k: np.array(v, dtype=float) for k, v in th.items()}
        t2["smooth"] = np.zeros(3)
        got = ca.cov_rns_taper(t2, locs, X, ci, rp, (nu, nu))
        want = oracle.cov_rns_taper(t2, locs, X, ci, rp, (nu, nu))
        assert _relerr(got, want) < ENTRY_RTOL
    m = 500
    lp = rng.uniform(0, 1, size=(m, 2))
    lp[10] = locs[77]
    lp[11] = np.array([5.0, 5.0])                        # no neighbour: an empty row
    Xp = np.column_stack([np.ones(m), rng.standard_normal(m), rng.standard_normal(m)])
    cip, rpp = _csr_within(lp, locs, 0.1)
    assert rpp[12] == rpp[11]
    got = ca.cov_rns_taper_pred(th, locs, lp, X, Xp, cip, rpp, wl_limits())
    want = oracle.cov_rns_taper_pred(th, locs, lp, X, Xp, cip, rpp, wl_limits())
    assert _relerr(got, want) < ENTRY_RTOL


def wl_limits():
    from cocons_amd import workloads as wl
    return wl.SMOOTH_LIMITS


def _wendland1(d, delta):
    """spam::cov.wend1 with range delta and sill 1: (1 - h)^4 (4 h + 1) for h = d / delta < 1 (the taper the
    package documentation uses for type = 'sparse')."""
    h = np.minimum(d / delta, 1.0)
    return (1.0 - h) ** 4 * (4.0 * h + 1.0)


def _taper_pattern(locs, delta):
    """(colindices, rowpointers, entries) of the Wendland-1 taper matrix, 1-based CSR as spam stores it."""
    ci, rp = _csr_within(locs, locs, delta)
    ent = np.empty(ci.size)
    for i in range(locs.shape[0]):
        w0, w1 = rp[i] - 1, rp[i + 1] - 1
        d = np.sqrt(np.sum((locs[ci[w0:w1] - 1] - locs[i]) ** 2, axis=1))
        ent[w0:w1] = _wendland1(d, delta)
    return ci, rp, ent


@pytest.mark.parametrize("n,r", [(150, 1), (700, 2), (1500, 1), (4000, 1)])
def test_taper_objective_vs_oracle(oracle, n, r):
    """GetNeg2loglikelihoodTaper / ...TaperProfile (R/neg2loglikelihood.R:20-108) on a taper handle: the value of
    the tapered covariance through the dense factorisation, against the CPU restatement (dense Cholesky of the same
    matrix), plain and engine schedule sizes, one and two realisations, with the penalty."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs, X, th, rng = _problem(n, seed=700 + n)
    z = rng.standard_normal((n, r))
    delta = 0.25 if n < 1000 else (0.12 if n < 3000 else 0.06)     # n = 4000: envelope of ~4 of 32 tile rows (band-limited path)
    ref_taper = _taper_pattern(locs, delta)
    pp = wl.par_pos_full()
    tv = wl.theta_vector_from_lists(th, pp)
    lam = (0.1, 0.2, 0.3)
    fit = ca.CoconsTaperFit(locs, X, z, wl.SMOOTH_LIMITS, *ref_taper)
    got = ca.GetNeg2loglikelihoodTaper(tv, pp, ref_taper, locs, X, wl.SMOOTH_LIMITS, z, n, lam, fit=fit)
    want = oracle.GetNeg2loglikelihoodTaper(tv, pp, ref_taper, locs, X, wl.SMOOTH_LIMITS, z, n, lam)
    assert abs(got - want) <= N2LL_RTOL * abs(want)
    gotp = ca.GetNeg2loglikelihoodTaperProfile(tv, pp, ref_taper, locs, X, wl.SMOOTH_LIMITS, z, n, lam, fit=fit)
    wantp = oracle.GetNeg2loglikelihoodTaperProfile(tv, pp, ref_taper, locs, X, wl.SMOOTH_LIMITS, z, n, lam)
    assert abs(gotp - wantp) <= N2LL_RTOL * abs(wantp)
    # a second evaluation on the same handle (the buffer is re-zeroed), and the handle-less form
    tv2 = tv.copy()
    tv2[0] += 0.05
    got2 = ca.GetNeg2loglikelihoodTaper(tv2, pp, ref_taper, locs, X, wl.SMOOTH_LIMITS, z, n, lam, fit=fit)
    want2 = oracle.GetNeg2loglikelihoodTaper(tv2, pp, ref_taper, locs, X, wl.SMOOTH_LIMITS, z, n, lam)
    assert abs(got2 - want2) <= N2LL_RTOL * abs(want2)
    if n <= 200:
        got3 = ca.GetNeg2loglikelihoodTaper(tv, pp, ref_taper, locs, X, wl.SMOOTH_LIMITS, z, n, lam)
        assert got3 == got
    # the batch entry (getHessian's points) runs them one after the other on a taper handle
    tl, tl2 = ca.getModelLists(tv, pp, "diff"), ca.getModelLists(tv2, pp, "diff")
    vals, st = fit.neg2loglik_batch_core([tl, tl2, tl])
    assert np.all(st == 0) and vals[0] == vals[2]
    pen, pen2 = ca.getPen(n * r, lam, tl, wl.SMOOTH_LIMITS), ca.getPen(n * r, lam, tl2, wl.SMOOTH_LIMITS)
    assert abs(vals[0] + pen - got) <= 1e-12 * abs(got) and abs(vals[1] + pen2 - got2) <= 1e-12 * abs(got2)
    # what a taper handle does not offer is refused, not computed on the wrong matrix
    with pytest.raises(ca.CoconsHipError, match="taper"):
        fit.cov_rows(th, np.array([0]))
    fit.close()


def test_taper_handle_recovers_after_failed_evaluation(oracle):
    """A band-limited taper handle must not depend on what an earlier evaluation left in the buffer: a NaN parameter and
    a non-positive-definite one (each poisons every tile the factorisation touches) followed by a valid one on the SAME
    handle -- what an optimiser does after GetNeg2loglikelihoodTaper(safe = TRUE) returned 1e6
    (R/neg2loglikelihood.R:33-38) -- and a prediction, which regrows the buffer, followed by the objective."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    n = 4000                                   # envelope of ~4 of 32 tile rows: odd tile columns exist inside it
    locs, X, th, rng = _problem(n, seed=4700)
    z = rng.standard_normal((n, 1))
    ref_taper = _taper_pattern(locs, 0.06)
    pp = wl.par_pos_full()
    tv = wl.theta_vector_from_lists(th, pp)
    lam = (0.0, 0.0, 0.0)
    fit = ca.CoconsTaperFit(locs, X, z, wl.SMOOTH_LIMITS, *ref_taper)
    want = oracle.GetNeg2loglikelihoodTaper(tv, pp, ref_taper, locs, X, wl.SMOOTH_LIMITS, z, n, lam)
    first = ca.GetNeg2loglikelihoodTaper(tv, pp, ref_taper, locs, X, wl.SMOOTH_LIMITS, z, n, lam, fit=fit)
    assert abs(first - want) <= N2LL_RTOL * abs(want)
    bad = tv.copy()
    bad[0] = np.nan
    assert ca.GetNeg2loglikelihoodTaper(bad, pp, ref_taper, locs, X, wl.SMOOTH_LIMITS, z, n, lam, fit=fit) == 1e6
    again = ca.GetNeg2loglikelihoodTaper(tv, pp, ref_taper, locs, X, wl.SMOOTH_LIMITS, z, n, lam, fit=fit)
    assert again == first
    # not positive definite: a huge negative nugget intercept cannot do it (nugget >= 0), a variance of -inf does:
    # std.dev -> exp(-inf) = 0 on the whole diagonal
    th_bad = {k: np.array(v, dtype=float).copy() for k, v in th.items()}
    th_bad["std.dev"][0] = -np.inf
    th_bad["nugget"][0] = -np.inf
    with pytest.raises(ca.CholeskyError):
        fit.neg2loglik_core(th_bad)
    again = ca.GetNeg2loglikelihoodTaper(tv, pp, ref_taper, locs, X, wl.SMOOTH_LIMITS, z, n, lam, fit=fit)
    assert again == first
    # a prediction with a large border reallocates the factorisation buffer; the objective afterwards is unchanged
    m = 300
    lp = rng.uniform(0, 1, size=(m, 2))
    Xp = np.column_stack([np.ones(m), rng.standard_normal(m), rng.standard_normal(m)])
    cip, rpp = _csr_within(lp, locs, 0.06)
    entp = np.ones(cip.size)
    fit.predict_core(th, lp, Xp, (cip, rpp, entp))
    assert ca.GetNeg2loglikelihoodTaper(bad, pp, ref_taper, locs, X, wl.SMOOTH_LIMITS, z, n, lam, fit=fit) == 1e6
    again = ca.GetNeg2loglikelihoodTaper(tv, pp, ref_taper, locs, X, wl.SMOOTH_LIMITS, z, n, lam, fit=fit)
    assert again == first
    fit.close()


def test_taper_packed_band_buffer_equals_dense_buffer(monkeypatch):
    """A band-limited taper handle keeps its factorisation buffer PACKED (every 128-column tile column stores its envelope
    rows and the rows under the matrix: O(n x bandwidth) doubles); COCONS_TAPER_PACKED=0 keeps the dense n x n buffer and
    only uses its band.  Same kernels, same order of operations: objective, parts and the sparse prediction are identical."""
    import cocons_amd as ca
    n, m = 3000, 200
    locs, X, th, rng = _problem(n, seed=3100)
    z = rng.standard_normal((n, 2))
    ref_taper = _taper_pattern(locs, 0.07)
    lp = rng.uniform(0, 1, size=(m, 2))
    Xp = np.column_stack([np.ones(m), rng.standard_normal(m), rng.standard_normal(m)])
    cip, rpp = _csr_within(lp, locs, 0.07)
    entp = np.empty(cip.size)
    for i in range(m):
        w0, w1 = rpp[i] - 1, rpp[i + 1] - 1
        d = np.sqrt(np.sum((locs[cip[w0:w1] - 1] - lp[i]) ** 2, axis=1))
        entp[w0:w1] = _wendland1(d, 0.07)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("COCONS_TAPER_PACKED", flag)
        fit = ca.CoconsTaperFit(locs, X, z, wl_limits(), *ref_taper)
        v, parts = fit.neg2loglik_core(th)
        st, qf = fit.predict_core(th, lp, Xp, (cip, rpp, entp))
        v2, _ = fit.neg2loglik_core(th)                      # after the prediction regrew the rows under the matrix
        res[flag] = (v, parts, st, qf, v2)
        fit.close()
    assert res["1"][0] == res["0"][0] and np.array_equal(res["1"][1], res["0"][1])
    assert np.array_equal(res["1"][2], res["0"][2]) and np.array_equal(res["1"][3], res["0"][3])
    assert res["1"][4] == res["1"][0] and res["0"][4] == res["0"][0]


def test_taper_predict_vs_oracle(oracle):
    """Sparse branch of cocoPredict (R/predict.R:216-283) on a taper handle against the CPU restatement: tapered
    cross-covariance rows as border of the tapered matrix, one prediction location without any neighbour (an empty
    row: stochastic 0, full marginal variance) and one on top of an observation."""
    import cocons_amd as ca
    n, m = 900, 300
    locs, X, th, rng = _problem(n, seed=77)
    z = rng.standard_normal(n)
    delta = 0.15
    ref_taper = _taper_pattern(locs, delta)
    lp = rng.uniform(0, 1, size=(m, 2))
    lp[3] = locs[100]
    lp[4] = np.array([4.0, 4.0])
    Xp = np.column_stack([np.ones(m), rng.standard_normal(m), rng.standard_normal(m)])
    cip, rpp = _csr_within(lp, locs, delta)
    entp = np.empty(cip.size)
    for i in range(m):
        w0, w1 = rpp[i] - 1, rpp[i + 1] - 1
        d = np.sqrt(np.sum((locs[cip[w0:w1] - 1] - lp[i]) ** 2, axis=1))
        entp[w0:w1] = _wendland1(d, delta)
    pred_taper = (cip, rpp, entp)
    assert rpp[5] == rpp[4]
    fit = ca.CoconsTaperFit(locs, X, z, wl_limits(), *ref_taper)
    got = ca.cocoPredict_sparse(th, locs, lp, X, Xp, wl_limits(), z, ref_taper, pred_taper, fit=fit)
    want = oracle.cocoPredict_sparse(th, locs, lp, X, Xp, wl_limits(), z, ref_taper, pred_taper)
    assert np.allclose(got["systematic"], want["systematic"], rtol=1e-13, atol=0)
    scale = np.max(np.abs(want["stochastic"]))
    assert np.max(np.abs(got["stochastic"] - want["stochastic"])) < 1e-10 * scale
    assert got["stochastic"][4] == 0.0
    assert np.max(np.abs(got["sd.pred"] - want["sd.pred"])) < 1e-9 * np.max(want["sd.pred"])
    # the objective still works on the handle afterwards (the prediction grew the buffer)
    v, _ = fit.neg2loglik_core(th)
    from cocons_amd import workloads as wl
    pp = wl.par_pos_full()
    tv = wl.theta_vector_from_lists(th, pp)
    w = oracle.GetNeg2loglikelihoodTaper(tv, pp, ref_taper, locs, X, wl_limits(), z.reshape(-1, 1), n, (0, 0, 0))
    assert abs(v - w) <= N2LL_RTOL * abs(w)
    fit.close()


def test_taper_fit_rejects_bad_patterns():
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs, X, th, rng = _problem(50, seed=5)
    ci, rp, ent = _taper_pattern(locs, 0.3)
    z = rng.standard_normal(50)
    with pytest.raises(ca.CoconsHipError, match="rowpointers"):
        ca.CoconsTaperFit(locs, X, z, wl.SMOOTH_LIMITS, ci, rp - 1, ent)          # 0-based pointers
    keep = np.ones(ci.size, dtype=bool)
    keep[rp[7] - 1 + int(np.nonzero(ci[rp[7] - 1:rp[8] - 1] == 8)[0][0])] = False   # drop the diagonal of row 8
    rp2 = rp.copy()
    rp2[8:] -= 1
    with pytest.raises(ca.CoconsHipError, match="diagonal"):
        ca.CoconsTaperFit(locs, X, z, wl.SMOOTH_LIMITS, ci[keep], rp2, ent[keep])


def test_cov_rows_and_cor_rows(oracle):
    """Rows of Sigma / cov2cor(Sigma) straight from the fit (cocons_cov_rows), the consumers of
    getCovMatrix (R/getFunctions.R:44-52, R/methods.R:161-165), against the full CPU matrix: both
    orientations around the requested index, the duplicate-location rule, classic and fixed-nu modes,
    on scattered (internally Morton-sorted) data."""
    import cocons_amd as ca
    n = 700
    locs, X, th, rng = _problem(n, seed=41)
    locs[5] = locs[400]                                     # u <= eps pair
    fit = ca.CoconsFit(locs, X, rng.standard_normal(n), wl_limits())
    idx = np.array([0, 5, 399, 400, n - 1])
    S = oracle.cov_rns(th, locs, X, wl_limits())
    rows = fit.cov_rows(th, idx)
    assert _relerr(rows, S[idx]) < ENTRY_RTOL
    d = 1 / np.sqrt(np.diag(S))
    R = (d[:, None] * S) * d[None, :]
    np.fill_diagonal(R, 1.0)
    cor = fit.cov_rows(th, idx, cor=True)
    assert _relerr(cor, R[idx]) < ENTRY_RTOL
    assert all(cor[b, i] == 1.0 for b, i in enumerate(idx))
    Sc = oracle.cov_rns_classic(th, locs, X)
    assert _relerr(fit.cov_rows(th, idx, classic=True), Sc[idx]) < ENTRY_RTOL
    t2 = {k: np.array(v, dtype=float) for k, v in th.items()}
    t2["smooth"] = np.zeros(3)
    fit2 = ca.CoconsFit(locs, X, rng.standard_normal(n), (1.5, 1.5))
    assert _relerr(fit2.cov_rows(t2, idx), oracle.cov_rns(t2, locs, X, (1.5, 1.5))[idx]) < ENTRY_RTOL


def test_device_matern_large_argument_branch_vs_mpmath():
    """The device routine's large-argument branch (Hankel series, u >= 20) and the CF2 branch next to it
    against 40-digit mpmath at 1500 random (nu, u), u in [15, 60] and [60, 700], nu in [0.25, 3.5]."""
    import mpmath as mp
    from cocons_amd import _lib
    mp.mp.dps = 40
    rng = np.random.default_rng(77)
    nu = rng.uniform(0.25, 3.5, 1500)
    u = np.concatenate([rng.uniform(15.0, 60.0, 1000), rng.uniform(60.0, 700.0, 500)])
    u[:6] = [19.999999, 20.0, 20.000001, 20.5, 24.0, 32.0]
    want = np.array([float(mp.power(2, 1 - mp.mpf(a)) / mp.gamma(mp.mpf(a)) * mp.power(mp.mpf(b), mp.mpf(a)) *
                           mp.besselk(mp.mpf(a), mp.mpf(b))) for a, b in zip(nu, u)])
    out = np.empty_like(u)
    L = _lib.load()
    _lib.check(L.cocons_debug_matern(u.size, nu.ctypes.data_as(_lib.c_dp), u.ctypes.data_as(_lib.c_dp),
                                     out.ctypes.data_as(_lib.c_dp)), "cocons_debug_matern")
    ok = want > 1e-290
    rel = np.abs(out[ok] - want[ok]) / want[ok]
    assert rel.max() < 1e-13, (rel.max(), nu[ok][rel.argmax()], u[ok][rel.argmax()])


def test_device_matern_middle_band_vs_mpmath():
    """The device routine's middle band, 2 <= u < 20 (round 4: the trapezoid rule on the integral representation, one
    exponential per node, replaces Steed's continued fraction for nu <= 3.5) against 40-digit mpmath on a DENSE grid:
    200 arguments across the band (both switch points and their neighbours included) x 15 orders -- tiny, half-integer,
    integer, just beside them, the largest order the branch takes and the first one it leaves to the continued fraction."""
    import mpmath as mp
    from cocons_amd import _lib
    mp.mp.dps = 40
    us = np.concatenate([np.linspace(2.0, 20.0, 194), [1.9999999, 2.0000001, 19.9999999, 3.14159, 7.75, 12.5]])
    nus = np.array([0.01, 0.1, 0.25, 0.5, 0.5000001, 0.9999999, 1.0, 1.3, 1.5, 2.0, 2.4999, 2.5, 3.0, 3.5, 3.5000001])
    nu = np.repeat(nus, us.size)
    u = np.tile(us, nus.size)
    want = np.array([float(mp.power(2, 1 - mp.mpf(a)) / mp.gamma(mp.mpf(a)) * mp.power(mp.mpf(b), mp.mpf(a)) *
                           mp.besselk(mp.mpf(a), mp.mpf(b))) for a, b in zip(nu, u)])
    out = np.empty_like(u)
    L = _lib.load()
    _lib.check(L.cocons_debug_matern(u.size, nu.ctypes.data_as(_lib.c_dp), u.ctypes.data_as(_lib.c_dp),
                                     out.ctypes.data_as(_lib.c_dp)), "cocons_debug_matern")
    rel = np.abs(out - want) / want
    assert rel.max() < 1e-13, (rel.max(), nu[rel.argmax()], u[rel.argmax()])


def test_device_matern_asymptotic_stand_in_vs_mpmath():
    """u >= 706: the reference replaces K_nu by its leading asymptotic term, 2^(1-nu)/Gamma(nu) u^nu sqrt(pi/2u) e^-u
    (src/cocons_full.cpp:301-305).  The device routine forms it with its own exponential and reciprocal-gamma table, not
    with pow / tgamma: against 40-digit mpmath where the value is still a normal double."""
    import mpmath as mp
    from cocons_amd import _lib
    mp.mp.dps = 40
    rng = np.random.default_rng(79)
    nu = np.concatenate([rng.uniform(0.25, 3.5, 200), rng.uniform(3.5, 12.0, 50)])
    u = np.concatenate([rng.uniform(706.0, 712.0, 240), [706.0, 706.0000001, 720.5, 730.0, 744.0, 760.0, 800.0, 1500.0, 900.0, 708.0]])
    want = np.array([float(mp.power(2, 1 - mp.mpf(a)) / mp.gamma(mp.mpf(a)) * mp.power(mp.mpf(b), mp.mpf(a)) *
                           mp.sqrt(mp.pi / (2 * mp.mpf(b))) * mp.exp(-mp.mpf(b))) for a, b in zip(nu, u)])
    out = np.empty_like(u)
    L = _lib.load()
    _lib.check(L.cocons_debug_matern(u.size, nu.ctypes.data_as(_lib.c_dp), u.ctypes.data_as(_lib.c_dp),
                                     out.ctypes.data_as(_lib.c_dp)), "cocons_debug_matern")
    ok = want > 1e-305                                          # comfortably normal doubles
    assert ok.sum() > 100
    rel = np.abs(out[ok] - want[ok]) / want[ok]
    assert rel.max() < 1e-12, (rel.max(), nu[ok][rel.argmax()], u[ok][rel.argmax()])
    assert np.all(np.isfinite(out)) and np.all(out[~ok] <= 1e-304)          # underflow: tiny or zero, never NaN


def test_device_matern_large_orders_vs_mpmath():
    """Orders beyond what the Hankel branch is validated for (nu > 3.5; smooth_limits are the user's) stay on the
    continued fraction for every u: against 40-digit mpmath at nu up to 15, u from 2 to 300."""
    import mpmath as mp
    from cocons_amd import _lib
    mp.mp.dps = 40
    rng = np.random.default_rng(78)
    nu = np.concatenate([rng.uniform(3.5, 15.0, 300), [3.5, 3.5000001, 8.0, 15.0, 15.0, 12.25]])
    u = np.concatenate([rng.uniform(2.0, 60.0, 200), rng.uniform(60.0, 300.0, 100), [20.0, 20.0, 20.0, 20.0, 45.0, 19.9]])
    want = np.array([float(mp.power(2, 1 - mp.mpf(a)) / mp.gamma(mp.mpf(a)) * mp.power(mp.mpf(b), mp.mpf(a)) *
                           mp.besselk(mp.mpf(a), mp.mpf(b))) for a, b in zip(nu, u)])
    out = np.empty_like(u)
    L = _lib.load()
    _lib.check(L.cocons_debug_matern(u.size, nu.ctypes.data_as(_lib.c_dp), u.ctypes.data_as(_lib.c_dp),
                                     out.ctypes.data_as(_lib.c_dp)), "cocons_debug_matern")
    ok = want > 1e-290
    rel = np.abs(out[ok] - want[ok]) / want[ok]
    assert rel.max() < 2e-13, (rel.max(), nu[ok][rel.argmax()], u[ok][rel.argmax()])


_FRESH_PROCESS_SCRIPT = r"""
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
import cocons_amd as ca
from cocons_amd import workloads as wl
rng = np.random.default_rng(11)
n = 700
locs = rng.uniform(0, 1, size=(n, 2))
X = wl.design_from_locs(locs)["std.covs"]
z = rng.standard_normal(n)
th = wl.theta_full(scale0=np.log(0.2))
fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
v = fit.neg2loglik_core(th)[0]          # the FIRST operation of this process, and it takes the engine schedule (nt = 6)
st = fit.engine_state()
fit.close()
print("RESULT " + json.dumps({"value": v, "state": st}))
"""


def test_first_engine_operation_of_a_fresh_process(record_property):
    """Round 3's driver run recorded a gate time-out (abort code unknown at the time) on the first engine-schedule operation
    of a fresh process, started while this process held a context on the same device: exactly that situation, once.  The
    child's first operation must run on the engine schedule without a time-out and give this process's value; the child's
    engine state (retries, abort code) goes into the test record either way."""
    import json
    import subprocess
    import sys
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    rng = np.random.default_rng(11)
    n = 700
    locs = rng.uniform(0, 1, size=(n, 2))
    X = wl.design_from_locs(locs)["std.covs"]
    z = rng.standard_normal(n)
    th = wl.theta_full(scale0=np.log(0.2))
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)          # this process holds a context and a handle meanwhile
    want = fit.neg2loglik_core(th)[0]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _FRESH_PROCESS_SCRIPT % {"root": root}], capture_output=True, text=True,
                         timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    record_property("fresh_process_engine_state", "retries=%d last_abort=0x%x active=%d"
                    % (res["state"]["retries"], res["state"]["last_abort"], int(res["state"]["active"])))
    assert res["state"]["retries"] == 0, res["state"]
    assert res["state"]["active"] == (os.environ.get("COCONS_ENGINE", "1") != "0"), res["state"]
    assert res["value"] == want                               # bit-identical: same kernels, same schedule
    fit.close()


def test_fit_same_data_is_a_bitwise_comparison():
    """cocons_fit_same_data (what the R glue's handle cache asks when its address check misses): 1 for the data the handle
    was created with -- at any address --, 0 as soon as one element, a dimension or smooth.limits differs."""
    import ctypes
    import cocons_amd as ca
    from cocons_amd import _lib, workloads as wl
    rng = np.random.default_rng(2)
    n = 300
    locs = np.asfortranarray(rng.uniform(0, 1, size=(n, 2)))
    X = np.asfortranarray(wl.design_from_locs(locs)["std.covs"])
    z = np.asfortranarray(rng.standard_normal((n, 2)))
    sl = np.asarray(wl.SMOOTH_LIMITS, dtype=np.float64)
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
    L = _lib.load()
    p = lambda a: a.ctypes.data_as(_lib.c_dp)

    def same(l, x, zz, s, nn=n, r=2):
        return L.cocons_fit_same_data(fit._h, nn, 3, r, 0, p(l), p(x), p(zz), None, p(s))
    assert same(locs, X, z, sl) == 1
    assert same(locs.copy(order="F"), X.copy(order="F"), z.copy(order="F"), sl.copy()) == 1
    z2 = z.copy(order="F"); z2[n - 1, 1] = np.nextafter(z2[n - 1, 1], 1e9)
    assert same(locs, X, z2, sl) == 0
    l2 = locs.copy(order="F"); l2[0, 0] += 1e-16 + abs(l2[0, 0]) * 1e-15
    assert same(l2, X, z, sl) == 0
    assert same(locs, X, z, np.array([0.5, 2.4])) == 0
    assert same(locs, X, z, sl, nn=n - 1) == 0
    assert same(locs, X, z, sl, r=1) == 0
    fit.close()
