"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's dense hot path above the C oracle
(`oracle/cocons_oracle.c`): the R closures of /root/reference/R/neg2loglikelihood.R,
the theta plumbing of R/getFunctions.R and the penalty of R/checkFunctions.R,
restated in numpy.  Cholesky / triangular solves go through scipy's LAPACK
(`dpotrf`, `dtrtrs`) -- the same LAPACK routines `base::chol` / `forwardsolve`
dispatch to (R's LAPACK is not vendored in the reference; version unpinned).

PARITY UNPINNED w.r.t. the reference binary (see the C file's header).  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from collections import OrderedDict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ASPECTS = ("mean", "std.dev", "scale", "aniso", "tilt", "smooth", "nugget")  # R/profile.R:5-7
COV_ASPECTS = ASPECTS[1:]


def build(force: bool = False) -> str:
    """Compile the C oracle in place (gcc, no GPU needed)."""
    so = os.path.join(_HERE, "libcocons_oracle.so")
    src = os.path.join(_HERE, "cocons_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        dp = ctypes.POINTER(ctypes.c_double)
        ci = ctypes.c_int
        L.oracle_besselk.restype = ctypes.c_double
        L.oracle_besselk.argtypes = [ctypes.c_double, ctypes.c_double]
        L.oracle_cov_rns.argtypes = [ci, ci, dp, dp, dp, dp, dp]
        L.oracle_cov_rns_classic.argtypes = [ci, ci, dp, dp, dp, dp]
        L.oracle_cov_rns_pred.argtypes = [ci, ci, ci, dp, dp, dp, dp, dp, dp, dp]
        L.oracle_sumsmoothlone.restype = ctypes.c_double
        L.oracle_sumsmoothlone.argtypes = [dp, ci, ctypes.c_double, ctypes.c_double]
        L.oracle_chol_ld.argtypes = [ci, dp, ci, dp, dp, dp, dp]
        ip = ctypes.POINTER(ctypes.c_int)
        L.oracle_cov_rns_taper.argtypes = [ci, ci, dp, dp, dp, dp, ci, ip, ip, dp]
        L.oracle_cov_rns_taper_pred.argtypes = [ci, ci, ci, dp, dp, dp, dp, dp, dp, ci, ip, ip, dp]
        _LIB = L
    return _LIB


def _f(a):
    """Column-major float64 copy (R layout)."""
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def theta_table(theta) -> np.ndarray:
    """named list (dict) of six length-p vectors -> 6 x p row-major table.
    Lookup is by name, like `theta["std.dev"]` in src/cocons_full.cpp:47-54, so a
    dict that also carries "mean" is accepted."""
    rows = [np.asarray(theta[k], dtype=np.float64).ravel() for k in COV_ASPECTS]
    return np.ascontiguousarray(np.stack(rows, axis=0))


def besselk(nu: float, x: float) -> float:
    return lib().oracle_besselk(float(nu), float(x))


def cov_rns(theta, locs, x_covariates, smooth_limits) -> np.ndarray:
    """src/cocons_full.cpp:40-321"""
    locs, X = _f(locs), _f(x_covariates)
    n, p = X.shape
    T = theta_table(theta)
    sl = np.asarray(smooth_limits, dtype=np.float64)
    out = np.empty((n, n), order="F")
    rc = lib().oracle_cov_rns(n, p, _p(T), _p(locs), _p(X), _p(sl), _p(out))
    assert rc == 0
    return out


def cov_rns_classic(theta, locs, x_covariates) -> np.ndarray:
    """src/cocons_full.cpp:480-594"""
    locs, X = _f(locs), _f(x_covariates)
    n, p = X.shape
    T = theta_table(theta)
    out = np.empty((n, n), order="F")
    rc = lib().oracle_cov_rns_classic(n, p, _p(T), _p(locs), _p(X), _p(out))
    assert rc == 0
    return out


def cov_rns_pred(theta, locs, locs_pred, x_covariates, x_covariates_pred, smooth_limits) -> np.ndarray:
    """src/cocons_full.cpp:334-471 -- returns m x n (row = prediction location)."""
    locs, lp, X, Xp = _f(locs), _f(locs_pred), _f(x_covariates), _f(x_covariates_pred)
    n, p = X.shape
    m = Xp.shape[0]
    T = theta_table(theta)
    sl = np.asarray(smooth_limits, dtype=np.float64)
    out = np.empty((m, n), order="F")
    rc = lib().oracle_cov_rns_pred(n, m, p, _p(T), _p(locs), _p(lp), _p(X), _p(Xp), _p(sl), _p(out))
    assert rc == 0
    return out


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def cov_rns_taper(theta, locs, x_covariates, colindices, rowpointers, smooth_limits) -> np.ndarray:
    """src/cocons_taper.cpp:151-433 -- the entries of the CSR pattern (colindices / rowpointers 1-based,
    as spam stores them; not modified)."""
    locs, X = _f(locs), _f(x_covariates)
    n, p = X.shape
    T = theta_table(theta)
    sl = np.asarray(smooth_limits, dtype=np.float64)
    ci = np.ascontiguousarray(colindices, dtype=np.int32)
    rp = np.ascontiguousarray(rowpointers, dtype=np.int32)
    out = np.empty(ci.size)
    rc = lib().oracle_cov_rns_taper(n, p, _p(T), _p(locs), _p(X), _p(sl), ci.size, _ip(ci), _ip(rp), _p(out))
    assert rc == 0, rc
    return out


def cov_rns_taper_pred(theta, locs, locs_pred, x_covariates, x_covariates_pred, colindices, rowpointers,
                       smooth_limits) -> np.ndarray:
    """src/cocons_taper.cpp:17-139 -- rows = prediction locations."""
    locs, lp, X, Xp = _f(locs), _f(locs_pred), _f(x_covariates), _f(x_covariates_pred)
    n, p = X.shape
    m = Xp.shape[0]
    T = theta_table(theta)
    sl = np.asarray(smooth_limits, dtype=np.float64)
    ci = np.ascontiguousarray(colindices, dtype=np.int32)
    rp = np.ascontiguousarray(rowpointers, dtype=np.int32)
    out = np.empty(ci.size)
    rc = lib().oracle_cov_rns_taper_pred(n, m, p, _p(T), _p(locs), _p(lp), _p(X), _p(Xp), _p(sl), ci.size,
                                         _ip(ci), _ip(rp), _p(out))
    assert rc == 0, rc
    return out


def sumsmoothlone(x, lam: float, alpha: float = 1e6) -> float:
    """src/cocons_full.cpp:12-30"""
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float64).ravel())
    return lib().oracle_sumsmoothlone(_p(x), x.size, float(lam), float(alpha))


# --------------------------------------------------------------------------- #
# theta plumbing (R/getFunctions.R)
# --------------------------------------------------------------------------- #
def _is_logical(v) -> bool:
    return isinstance(v, (list, tuple, np.ndarray)) and len(v) > 0 and \
        all(isinstance(b, (bool, np.bool_)) for b in v)


def getModelLists(theta, par_pos, type="diff"):
    """R/getFunctions.R:570-616.  `par_pos`: ordered mapping aspect -> list of bool
    (free columns) or a number (aspect fixed at that value in slot 1)."""
    theta = np.asarray(theta, dtype=np.float64).ravel()
    length_logical = max(len(v) if _is_logical(v) else 1 for v in par_pos.values())
    out = OrderedDict()
    acum = 0
    for name, pp in par_pos.items():
        vec = np.zeros(length_logical)
        if not _is_logical(pp):
            vec[0] = float(np.asarray(pp, dtype=np.float64).ravel()[0])   # :580-584
        else:
            mask = np.asarray(pp, dtype=bool)
            k = int(mask.sum())
            full = np.zeros(len(mask))
            full[mask] = theta[acum:acum + k]
            vec[:len(mask)] = full
            acum += k
        out[name] = vec
    if type == "classic":
        return out
    sd_pp, sc_pp = par_pos["std.dev"], par_pos["scale"]
    if _is_logical(sd_pp) and _is_logical(sc_pp):                         # :603-612
        tmp = OrderedDict((k, v.copy()) for k, v in out.items())
        for i in range(len(sd_pp)):
            if sd_pp[i] and sc_pp[i]:
                tmp["std.dev"][i] = (out["std.dev"][i] + out["scale"][i]) / 2
                tmp["scale"][i] = (out["std.dev"][i] - out["scale"][i]) / 2
        return tmp
    return out


def getScale(x, mean_vector=None, sd_vector=None):
    """R/getFunctions.R:410-434 (matrix branch): column 1 untouched, the others
    (x - mean) / sd with stats::sd (n-1 denominator)."""
    x = np.array(x, dtype=np.float64, copy=True, order="F")
    if mean_vector is None:
        mean_vector = x.mean(axis=0)
        mean_vector[0] = 0.0
    if sd_vector is None:
        sd_vector = x.std(axis=0, ddof=1) if x.shape[0] > 1 else np.ones(x.shape[1])
        sd_vector[0] = 1.0
    for ii in range(1, x.shape[1]):
        x[:, ii] = (x[:, ii] - mean_vector[ii]) / sd_vector[ii]
    return {"std.covs": x, "mean.vector": np.asarray(mean_vector), "sd.vector": np.asarray(sd_vector)}


def getPen(n, lam, theta_list, smooth_limits) -> float:
    """.cocons.getPen, R/checkFunctions.R:474-492 (lambda = Sigma, betas, reg)."""
    names = list(theta_list.keys())
    summ = lam[2] * math.exp(theta_list["scale"][0]) * math.sqrt(
        (smooth_limits[1] - smooth_limits[0]) / (1 + math.exp(-theta_list["smooth"][0])) + smooth_limits[0]
    ) + sumsmoothlone(theta_list[names[0]][1:], lam[1])
    for ii in range(1, 6):                       # R's 2:6 -> std.dev .. smooth
        summ += sumsmoothlone(theta_list[names[ii]][1:], lam[0])
    return 2 * n * summ


# --------------------------------------------------------------------------- #
# -2 log-likelihood objectives (R/neg2loglikelihood.R)
# --------------------------------------------------------------------------- #
def _chol_upper(S):
    from scipy.linalg import lapack
    R, info = lapack.dpotrf(S, lower=0, clean=1, overwrite_a=0)   # base::chol -> dpotrf('U')
    # A NaN pivot is a failure too: the reference LAPACK R ships (dpotrf -> dpotrf2 / dpotf2) tests
    # `AJJ.LE.ZERO .OR. DISNAN(AJJ)` at every pivot, so chol() of a matrix poisoned by NaN raises the error that
    # R/neg2loglikelihood.R:200-206 maps to 1e6; scipy's OpenBLAS dpotrf has its own kernels, which let NaN through with
    # info = 0.  (A NaN anywhere in the part of Sigma that is read reaches a later pivot.)
    if info == 0 and not np.all(np.isfinite(np.diag(R))):
        info = int(np.argmax(~np.isfinite(np.diag(R)))) + 1
    return (None if info != 0 else R), info


def _forwardsolve_t(R, b):
    """forwardsolve(R, b, transpose=TRUE, upper.tri=TRUE): solves R^T y = b."""
    from scipy.linalg import solve_triangular
    return solve_triangular(R, b, trans="T", lower=False, check_finite=False)


def _backsolve(R, b):
    from scipy.linalg import solve_triangular
    return solve_triangular(R, b, trans="N", lower=False, check_finite=False)


def GetNeg2loglikelihood(theta, par_pos, locs, x_covariates, smooth_limits, z, n, lam, safe=True):
    """R/neg2loglikelihood.R:183-222"""
    tl = getModelLists(theta, par_pos, "diff")
    Sigma = cov_rns(tl, locs, x_covariates, smooth_limits)
    R, info = _chol_upper(Sigma)
    if R is None:
        if safe:
            return 1e6
        raise RuntimeError("Cholesky error")
    logdet = float(np.sum(np.log(np.diag(R))))
    X = np.asarray(x_covariates, dtype=np.float64)
    z = np.asarray(z, dtype=np.float64).reshape(X.shape[0], -1)
    trend = X @ tl["mean"]
    total = 0.0
    for k in range(z.shape[1]):
        y = _forwardsolve_t(R, z[:, k] - trend)
        total += n * math.log(2 * math.pi) + 2 * logdet + float(y @ y)
    return total + getPen(n * z.shape[1], lam, tl, smooth_limits)


def _taper_dense(ref_taper, entries, n):
    """Dense symmetric matrix of a spam pattern (colindices, rowpointers 1-based) with the given entries."""
    ci, rp, _ = ref_taper
    S = np.zeros((n, n))
    for i in range(n):
        for w in range(rp[i] - 1, rp[i + 1] - 1):
            S[i, ci[w] - 1] = entries[w]
    return S


def GetNeg2loglikelihoodTaper(theta, par_pos, ref_taper, locs, x_covariates, smooth_limits, z, n, lam, safe=True):
    """R/neg2loglikelihood.R:20-53 with the sparse Cholesky (spam, absent here) replaced by a dense one of the
    same matrix: ref_taper@entries * cov_rns_taper(...) (:25-31), 2 * log det of the factor (:43) and
    crossprod(forwardsolve(cholS, resid)) (:49-50) are properties of the matrix, not of its storage.
    ref_taper = (colindices, rowpointers, entries), 1-based as spam stores them."""
    tl = getModelLists(theta, par_pos, "diff")
    ci, rp, te = ref_taper
    ent = np.asarray(te, dtype=np.float64) * cov_rns_taper(tl, locs, x_covariates, ci, rp, smooth_limits)
    R, info = _chol_upper(_taper_dense(ref_taper, ent, n))
    if R is None:
        if safe:
            return 1e6
        raise RuntimeError("Cholesky error")
    logdet = float(np.sum(np.log(np.diag(R))))
    X = np.asarray(x_covariates, dtype=np.float64)
    z = np.asarray(z, dtype=np.float64).reshape(X.shape[0], -1)
    trend = X @ tl["mean"]
    total = 0.0
    for k in range(z.shape[1]):
        y = _forwardsolve_t(R, z[:, k] - trend)
        total += n * math.log(2 * math.pi) + 2 * logdet + float(y @ y)
    return total + getPen(n * z.shape[1], lam, tl, smooth_limits)


def GetNeg2loglikelihoodTaperProfile(theta, par_pos, ref_taper, locs, x_covariates, smooth_limits, z, n, lam,
                                     safe=True):
    """R/neg2loglikelihood.R:73-108 (dense factorisation as above)."""
    tl = getModelLists(theta, par_pos, "diff")
    sd = np.array(tl["std.dev"], dtype=np.float64, copy=True)
    sd[0] = 0.0
    tl["std.dev"] = sd
    ci, rp, te = ref_taper
    ent = np.asarray(te, dtype=np.float64) * cov_rns_taper(tl, locs, x_covariates, ci, rp, smooth_limits)
    R, info = _chol_upper(_taper_dense(ref_taper, ent, n))
    if R is None:
        if safe:
            return 1e6
        raise RuntimeError("Cholesky error")
    logdet = float(np.sum(np.log(np.diag(R))))
    X = np.asarray(x_covariates, dtype=np.float64)
    z = np.asarray(z, dtype=np.float64).reshape(X.shape[0], -1)
    r = z.shape[1]
    trend = X @ tl["mean"]
    sum_in = 0.0
    for k in range(r):
        y = _forwardsolve_t(R, z[:, k] - trend)
        sum_in += float(y @ y)
    return (r * n * math.log(2 * math.pi) + r * n + r * 2 * logdet + r * n * math.log(sum_in / (r * n))
            + getPen(n * r, lam, tl, smooth_limits))


def GetNeg2loglikelihoodProfile(theta, par_pos, locs, x_covariates, smooth_limits, z, n, x_betas, lam, safe=True):
    """R/neg2loglikelihood.R:127-165 (literal: chol2inv + P_mat)."""
    tl = getModelLists(theta, par_pos, "diff")
    Sigma = cov_rns(tl, locs, x_covariates, smooth_limits)
    R, info = _chol_upper(Sigma)
    if R is None:
        if safe:
            return 1e6
        raise RuntimeError("Cholesky error")
    Xb = np.asarray(x_betas, dtype=np.float64)
    V = _backsolve(R, _forwardsolve_t(R, Xb))
    W = Xb.T @ V
    Sinv = _backsolve(R, _forwardsolve_t(R, np.eye(R.shape[0])))          # chol2inv
    P = Sinv - V @ np.linalg.solve(W, V.T)
    logdet = float(np.sum(np.log(np.diag(R))))
    z = np.asarray(z, dtype=np.float64).reshape(Xb.shape[0], -1)
    total = 0.0
    for k in range(z.shape[1]):
        total += n * math.log(2 * math.pi) + 2 * logdet + float(z[:, k] @ (P @ z[:, k]))
    return total + getPen(n * z.shape[1], lam, tl, smooth_limits)


def GetNeg2loglikelihoodREML(theta, par_pos, locs, x_covariates, x_betas, smooth_limits, z, n, lam, safe=True):
    """R/neg2loglikelihood.R:241-291 (note: V, W are built from x_covariates, :273-276)."""
    tl = getModelLists(theta, par_pos, "diff")
    Sigma = cov_rns(tl, locs, x_covariates, smooth_limits)
    R, info = _chol_upper(Sigma)
    if R is None:
        if safe:
            return 1e6
        raise RuntimeError("Cholesky error")
    X = np.asarray(x_covariates, dtype=np.float64)
    logdet = float(np.sum(np.log(np.diag(R))))
    p = int(np.linalg.matrix_rank(X))                                     # qr(x)$rank
    V = _backsolve(R, _forwardsolve_t(R, X))
    W = X.T @ V
    Sinv = _backsolve(R, _forwardsolve_t(R, np.eye(R.shape[0])))
    P = Sinv - V @ np.linalg.solve(W, V.T)
    cholW = np.linalg.cholesky(W)
    z = np.asarray(z, dtype=np.float64).reshape(X.shape[0], -1)
    total = 0.0
    for k in range(z.shape[1]):
        total += (n - p) * math.log(2 * math.pi) + 2 * logdet + \
            2 * float(np.sum(np.log(np.diag(cholW)))) + float(z[:, k] @ (P @ z[:, k]))
    return total + getPen((n - p) * z.shape[1], lam, tl, smooth_limits)


def getHessian_dense(par, par_pos, locs, x_covariates, smooth_limits, z, n, lam, f00=None,
                      eps=np.finfo(float).eps ** 0.25):
    """R/getFunctions.R:925-1034 (dense branch), serial."""
    par = np.asarray(par, dtype=np.float64).ravel()
    P = par.size

    def fn(t):
        return GetNeg2loglikelihood(t, par_pos, locs, x_covariates, smooth_limits, z, n, lam)

    if f00 is None:
        f00 = fn(par)
    H = np.zeros((P, P))
    for jj in range(P):
        for ii in range(jj, P):
            t01, t10, t11 = par.copy(), par.copy(), par.copy()
            t01[jj] += eps
            t10[ii] += eps
            t11[jj] += eps
            t11[ii] += eps
            H[jj, ii] = 0.5 * ((fn(t11) - fn(t01) - fn(t10) + f00) / (eps * eps))
    H = H + H.T
    H[np.diag_indices(P)] /= 2
    return H


def cocoPredict_dense(theta_list, locs, newlocs, X_std, X_pred_std, smooth_limits, z, type="pred"):
    """Dense branch of cocoPredict, R/predict.R:136-187, from the point where the
    scaled design matrices and the adjusted theta list exist."""
    observed_cov = cov_rns(theta_list, locs, X_std, smooth_limits)
    cov_pred = cov_rns_pred(theta_list, locs, newlocs, X_std, X_pred_std, smooth_limits)
    inv_cov = np.linalg.solve(observed_cov, cov_pred.T)                   # solve() -> dgesv
    Xs, Xp = np.asarray(X_std, float), np.asarray(X_pred_std, float)
    systematic_pred = Xp @ theta_list["mean"]
    resid = np.asarray(z, float).ravel() - Xs @ theta_list["mean"]
    stochastic = resid @ inv_cov
    if type == "mean":
        return {"systematic": systematic_pred, "stochastic": stochastic}
    unc = 1 / np.exp(-(Xp @ theta_list["std.dev"])) + np.exp(Xp @ theta_list["nugget"])
    unc = unc - np.sum(cov_pred * inv_cov.T, axis=1)
    neg = unc < 1e-10
    unc[neg] = np.abs(unc[neg])
    return {"systematic": systematic_pred, "stochastic": stochastic, "sd.pred": np.sqrt(unc)}


def cocoPredict_sparse(theta_list, locs, newlocs, X_std, X_pred_std, smooth_limits, z, ref_taper, pred_taper,
                       type="pred"):
    """Sparse branch of cocoPredict, R/predict.R:216-283, with spam::solve (absent here) replaced by a dense solve
    of the same matrices.  ref_taper / pred_taper = (colindices, rowpointers, entries), 1-based."""
    Xs, Xp = np.asarray(X_std, float), np.asarray(X_pred_std, float)
    n, m = Xs.shape[0], Xp.shape[0]
    ent = np.asarray(ref_taper[2], float) * cov_rns_taper(theta_list, locs, X_std, ref_taper[0], ref_taper[1], smooth_limits)
    S = _taper_dense(ref_taper, ent, n)
    entp = np.asarray(pred_taper[2], float) * cov_rns_taper_pred(theta_list, locs, newlocs, X_std, X_pred_std,
                                                                  pred_taper[0], pred_taper[1], smooth_limits)
    C = np.zeros((m, n))
    ci, rp = pred_taper[0], pred_taper[1]
    for i in range(m):
        for w in range(rp[i] - 1, rp[i + 1] - 1):
            C[i, ci[w] - 1] = entp[w]
    inv_cov = np.linalg.solve(S, C.T)                                            # :244
    systematic_pred = Xp @ theta_list["mean"]
    resid = np.asarray(z, float).ravel() - Xs @ theta_list["mean"]
    stochastic = resid @ inv_cov                                                 # :252
    if type == "mean":
        return {"systematic": systematic_pred, "stochastic": stochastic}
    unc = 1 / np.exp(-(Xp @ theta_list["std.dev"])) + np.exp(Xp @ theta_list["nugget"])
    unc = unc - np.sum(C * inv_cov.T, axis=1)                                    # :267
    neg = unc < 1e-10
    unc[neg] = np.abs(unc[neg])
    return {"systematic": systematic_pred, "stochastic": stochastic, "sd.pred": np.sqrt(unc)}


def cocoSim_dense(theta_list, locs, X_std, smooth_limits, iiderrors, type="classic"):
    """Marginal branch of cocoSim (dense), R/sim.R:147-172."""
    if type == "classic":
        covmat = cov_rns_classic(theta_list, locs, X_std)
    else:
        covmat = cov_rns(theta_list, locs, X_std, smooth_limits)
    R, info = _chol_upper(covmat)
    if R is None:
        raise RuntimeError("Cholesky error")
    E = np.asarray(iiderrors, dtype=np.float64).reshape(covmat.shape[0], -1)
    mu = np.asarray(X_std, float) @ np.asarray(theta_list["mean"], float)
    return (E.T @ R + mu[None, :]).T


def cocoSim_cond_dense(theta_list, locs, newlocs, newdataset, X_std, X_pred_std, smooth_limits, z, iiderrors):
    """Conditional branch of cocoSim (dense), R/sim.R:84-127, literal."""
    covmat = cov_rns(theta_list, locs, X_std, smooth_limits)
    covmat_pred = cov_rns_pred(theta_list, locs, newlocs, X_std, X_pred_std, smooth_limits)
    covmat_unobs = cov_rns(theta_list, np.asarray(newdataset, float)[:, :2], X_pred_std, smooth_limits)
    S = covmat_unobs - covmat_pred @ np.linalg.solve(covmat, covmat_pred.T)
    S = (S + S.T) / 2
    L, info = _chol_upper(np.asfortranarray(S))
    if L is None:
        raise RuntimeError("Cholesky error")
    step_one = cocoPredict_dense(theta_list, locs, newlocs, X_std, X_pred_std, smooth_limits, z, type="mean")
    tmp_mu = step_one["systematic"] + step_one["stochastic"]
    E = np.asarray(iiderrors, float).reshape(covmat_unobs.shape[0], -1)
    return (E.T @ L + tmp_mu[None, :]).T


def chol_ld(A, rhs):
    """long-double Cholesky truth: returns (info, sum(log(diag)), quad[nrhs], Y)."""
    A = _f(A)
    n = A.shape[0]
    rhs = _f(np.asarray(rhs, float).reshape(n, -1))
    k = rhs.shape[1]
    ld = ctypes.c_double(0.0)
    quad = np.zeros(k)
    Y = np.zeros((n, k), order="F")
    info = lib().oracle_chol_ld(n, _p(A), k, _p(rhs), ctypes.byref(ld), _p(quad), _p(Y))
    return info, ld.value, quad, Y
