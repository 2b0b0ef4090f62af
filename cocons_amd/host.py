"""Host-side mirror of the reference's R interface for the dense hot path.

The reference's host language is R (absent from this image), so the host side
above the C ABI is written in Python with the reference's names, argument
meaning and error behaviour:

  cov_rns / cov_rns_classic / cov_rns_pred     R/RcppExports.R:21-46
  getModelLists / getScale                     R/getFunctions.R:570-616, :376-436
  sumsmoothlone / getPen (.cocons.getPen)      src/cocons_full.cpp:12-30, R/checkFunctions.R:474-492
  GetNeg2loglikelihood[Profile|REML]           R/neg2loglikelihood.R:183-222, :127-165, :241-291
  cocoPredict_dense                            R/predict.R:136-187 (dense branch)

All heavy arithmetic runs in the HIP library; this module only does the O(p)
theta plumbing the reference also keeps on the host.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes
import math
from collections import OrderedDict

import numpy as np

from . import _lib
from ._lib import CholeskyError, c_dp

ASPECTS = ("mean", "std.dev", "scale", "aniso", "tilt", "smooth", "nugget")   # R/profile.R:5-7
COV_ASPECTS = ASPECTS[1:]


def _f(a):
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(c_dp)


def theta_table(theta) -> np.ndarray:
    """Named list of length-p vectors -> the 6 x p row-major table of the C ABI.
    Looked up by name like src/cocons_full.cpp:47-54, so `theta_list` and
    `theta_list[-1]` both work."""
    rows = [np.asarray(theta[k], dtype=np.float64).ravel() for k in COV_ASPECTS]
    p = rows[0].size
    if any(r.size != p for r in rows):
        raise ValueError("theta aspects must have equal length")
    if p > _lib.P_MAX:
        raise ValueError("design matrix has %d columns; the HIP path supports up to %d" % (p, _lib.P_MAX))
    return np.ascontiguousarray(np.stack(rows, axis=0))


# --------------------------------------------------------------------------- #
# covariance assembly (.Call surface)
# --------------------------------------------------------------------------- #
def cov_rns(theta, locs, x_covariates, smooth_limits) -> np.ndarray:
    """Dense covariance function (difference parameterization); R/RcppExports.R:21-23."""
    L = _lib.load()
    locs, X = _f(locs), _f(x_covariates)
    n, p = X.shape
    T = theta_table(theta)
    sl = np.asarray(smooth_limits, dtype=np.float64)
    out = np.empty((n, n), order="F")
    _lib.check(L.cocons_cov_rns(n, p, _p(T), _p(locs), _p(X), _p(sl), _p(out)), "cov_rns")
    return out


def cov_rns_classic(theta, locs, x_covariates) -> np.ndarray:
    """Dense covariance function (classic parameterization); R/RcppExports.R:44-46."""
    L = _lib.load()
    locs, X = _f(locs), _f(x_covariates)
    n, p = X.shape
    T = theta_table(theta)
    out = np.empty((n, n), order="F")
    _lib.check(L.cocons_cov_rns_classic(n, p, _p(T), _p(locs), _p(X), _p(out)), "cov_rns_classic")
    return out


def cov_rns_pred(theta, locs, locs_pred, x_covariates, x_covariates_pred, smooth_limits) -> np.ndarray:
    """Cross-covariance, m x n with row = prediction location; R/RcppExports.R:34-36."""
    L = _lib.load()
    locs, lp, X, Xp = _f(locs), _f(locs_pred), _f(x_covariates), _f(x_covariates_pred)
    n, p = X.shape
    m = Xp.shape[0]
    T = theta_table(theta)
    sl = np.asarray(smooth_limits, dtype=np.float64)
    out = np.empty((m, n), order="F")
    _lib.check(L.cocons_cov_rns_pred(n, m, p, _p(T), _p(locs), _p(lp), _p(X), _p(Xp), _p(sl), _p(out)),
               "cov_rns_pred")
    return out


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def cov_rns_taper(theta, locs, x_covariates, colindices, rowpointers, smooth_limits) -> np.ndarray:
    """Sparse covariance function: the entries of the spam pattern (colindices / rowpointers 1-based, as
    spam stores them); R/RcppExports.R:63-65 -> src/cocons_taper.cpp:151-433."""
    L = _lib.load()
    locs, X = _f(locs), _f(x_covariates)
    n, p = X.shape
    T = theta_table(theta)
    sl = np.asarray(smooth_limits, dtype=np.float64)
    ci = np.ascontiguousarray(colindices, dtype=np.int32)
    rp = np.ascontiguousarray(rowpointers, dtype=np.int32)
    out = np.empty(ci.size)
    _lib.check(L.cocons_cov_rns_taper(n, p, _p(T), _p(locs), _p(X), _p(sl), ci.size, _ip(ci), _ip(rp), _p(out)),
               "cov_rns_taper")
    return out


def cov_rns_taper_pred(theta, locs, locs_pred, x_covariates, x_covariates_pred, colindices, rowpointers,
                       smooth_limits) -> np.ndarray:
    """Sparse cross-covariance entries (rows = prediction locations); R/RcppExports.R:52-54 ->
    src/cocons_taper.cpp:17-139."""
    L = _lib.load()
    locs, lp, X, Xp = _f(locs), _f(locs_pred), _f(x_covariates), _f(x_covariates_pred)
    n, p = X.shape
    m = Xp.shape[0]
    T = theta_table(theta)
    sl = np.asarray(smooth_limits, dtype=np.float64)
    ci = np.ascontiguousarray(colindices, dtype=np.int32)
    rp = np.ascontiguousarray(rowpointers, dtype=np.int32)
    out = np.empty(ci.size)
    _lib.check(L.cocons_cov_rns_taper_pred(n, m, p, _p(T), _p(locs), _p(lp), _p(X), _p(Xp), _p(sl), ci.size,
                                           _ip(ci), _ip(rp), _p(out)), "cov_rns_taper_pred")
    return out


def sumsmoothlone(x, lam: float, alpha: float = 1e6) -> float:
    """Smoothed-L1 penalty; R/RcppExports.R:10-12 (host arithmetic, O(p))."""
    L = _lib.load()
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float64).ravel())
    return L.cocons_sumsmoothlone(_p(x), x.size, float(lam), float(alpha))


# --------------------------------------------------------------------------- #
# theta plumbing -- stays on the host in the reference too
# --------------------------------------------------------------------------- #
def _is_logical(v) -> bool:
    return isinstance(v, (list, tuple, np.ndarray)) and len(v) > 0 and \
        all(isinstance(b, (bool, np.bool_)) for b in v)


def getModelLists(theta, par_pos, type="diff"):
    """R/getFunctions.R:570-616."""
    theta = np.asarray(theta, dtype=np.float64).ravel()
    length_logical = max(len(v) if _is_logical(v) else 1 for v in par_pos.values())
    out = OrderedDict()
    acum = 0
    for name, pp in par_pos.items():
        vec = np.zeros(length_logical)
        if not _is_logical(pp):
            vec[0] = float(np.asarray(pp, dtype=np.float64).ravel()[0])
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
    if type != "diff":
        raise ValueError("type must be 'diff' or 'classic'")
    sd_pp, sc_pp = par_pos["std.dev"], par_pos["scale"]
    if _is_logical(sd_pp) and _is_logical(sc_pp):
        tmp = OrderedDict((k, v.copy()) for k, v in out.items())
        for i in range(len(sd_pp)):
            if sd_pp[i] and sc_pp[i]:
                tmp["std.dev"][i] = (out["std.dev"][i] + out["scale"][i]) / 2
                tmp["scale"][i] = (out["std.dev"][i] - out["scale"][i]) / 2
        return tmp
    return out


def getScale(x, mean_vector=None, sd_vector=None):
    """R/getFunctions.R:410-434 (matrix branch)."""
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
    """.cocons.getPen; R/checkFunctions.R:474-492 (lambda = Sigma, betas, reg)."""
    names = list(theta_list.keys())
    summ = lam[2] * math.exp(theta_list["scale"][0]) * math.sqrt(
        (smooth_limits[1] - smooth_limits[0]) / (1 + math.exp(-theta_list["smooth"][0])) + smooth_limits[0]
    ) + sumsmoothlone(theta_list[names[0]][1:], lam[1])
    for ii in range(1, 6):
        summ += sumsmoothlone(theta_list[names[ii]][1:], lam[0])
    return 2 * n * summ


# --------------------------------------------------------------------------- #
# fit handle: data that is constant over an optimisation stays in HBM
# --------------------------------------------------------------------------- #
class CoconsFit:
    """Device-resident (locs, x_covariates, z [, x_betas], smooth.limits) of one fit --
    the arguments the reference passes unchanged to every objective evaluation
    (R/optim.R:237-259).  Only O(p) bytes cross PCIe per evaluation."""

    def __init__(self, locs, x_covariates, z, smooth_limits, x_betas=None, device=-1):
        L = _lib.load()
        self._L = L
        locs, X = _f(locs), _f(x_covariates)
        self.n, self.p = X.shape
        if locs.shape != (self.n, 2):
            raise ValueError("locs must be n x 2")
        z = _f(np.asarray(z, dtype=np.float64).reshape(self.n, -1))
        self.r = z.shape[1]
        xb = None
        self.q = 0
        if x_betas is not None:
            xb = _f(np.asarray(x_betas, dtype=np.float64).reshape(self.n, -1))
            self.q = xb.shape[1]
        self.smooth_limits = np.asarray(smooth_limits, dtype=np.float64).copy()
        self.x_covariates = X
        self._h = L.cocons_fit_create(self.n, self.p, self.r, self.q, _p(locs), _p(X), _p(z),
                                      _p(xb) if xb is not None else None, _p(self.smooth_limits), int(device))
        if not self._h:
            raise _lib.CoconsHipError("cocons_fit_create failed: " + _lib.last_error())

    def close(self):
        if getattr(self, "_h", None):
            self._L.cocons_fit_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- cores (no penalty) ---------------------------------------------------
    def neg2loglik_core(self, theta_list):
        T = theta_table(theta_list)
        mean = np.ascontiguousarray(np.asarray(theta_list["mean"], dtype=np.float64))
        val = ctypes.c_double(0.0)
        parts = np.zeros(1 + self.r)
        _lib.check(self._L.cocons_neg2loglik_dense(self._h, _p(T), _p(mean), ctypes.byref(val), _p(parts)),
                   "cocons_neg2loglik_dense")
        return val.value, parts

    def neg2loglik_batch_core(self, theta_lists):
        """Independent evaluations pipelined on the GPU (cocons_neg2loglik_batch).  Returns
        (values, status) arrays; status k > 0 marks a Cholesky failure at minor k."""
        nb = len(theta_lists)
        T = np.ascontiguousarray(np.stack([theta_table(t) for t in theta_lists], axis=0)) if nb else np.zeros((0, 6, self.p))
        M = np.ascontiguousarray(np.stack([np.asarray(t["mean"], dtype=np.float64) for t in theta_lists], axis=0)) \
            if nb else np.zeros((0, self.p))
        vals = np.zeros(nb)
        st = np.zeros(nb, dtype=np.int32)
        _lib.check(self._L.cocons_neg2loglik_batch(self._h, nb, _p(T), _p(M), _p(vals),
                                                   st.ctypes.data_as(ctypes.POINTER(ctypes.c_int))),
                   "cocons_neg2loglik_batch")
        return vals, st

    def neg2loglik_profile_core(self, theta_list):
        T = theta_table(theta_list)
        val = ctypes.c_double(0.0)
        parts = np.zeros(2 + self.r + self.q)
        _lib.check(self._L.cocons_neg2loglik_profile(self._h, _p(T), ctypes.byref(val), _p(parts)),
                   "cocons_neg2loglik_profile")
        return val.value, parts

    def neg2loglik_reml_core(self, theta_list, rank):
        T = theta_table(theta_list)
        val = ctypes.c_double(0.0)
        parts = np.zeros(2 + self.r + self.p)
        _lib.check(self._L.cocons_neg2loglik_reml(self._h, _p(T), int(rank), ctypes.byref(val), _p(parts)),
                   "cocons_neg2loglik_reml")
        return val.value, parts

    def predict_core(self, theta_list, locs_pred, x_covariates_pred, z_col=0):
        T = theta_table(theta_list)
        mean = np.ascontiguousarray(np.asarray(theta_list["mean"], dtype=np.float64))
        lp, Xp = _f(locs_pred), _f(x_covariates_pred)
        m = Xp.shape[0]
        st, qf = np.zeros(m), np.zeros(m)
        _lib.check(self._L.cocons_predict_dense(self._h, _p(T), _p(mean), int(z_col), m, _p(lp), _p(Xp),
                                                _p(st), _p(qf)), "cocons_predict_dense")
        return st, qf

    def cov_rows(self, theta_list, index, cor=False, classic=False):
        """Rows `index` (0-based) of cov_rns / cov_rns_classic at the fit's locations, or of cov2cor of it,
        without the n x n matrix (what plot(type = "correlations") reads, R/methods.R:161-165)."""
        T = theta_table(theta_list)
        idx = np.ascontiguousarray(np.atleast_1d(index), dtype=np.int32)
        out = np.empty((idx.size, self.n))
        _lib.check(self._L.cocons_cov_rows(self._h, _p(T), int(bool(classic)), idx.size, _ip(idx), int(bool(cor)), _p(out)),
                   "cocons_cov_rows")
        return out

    def sim_core(self, theta_list, iiderrors, classic=False):
        E = _f(np.asarray(iiderrors, dtype=np.float64).reshape(self.n, -1))
        T = theta_table(theta_list)
        mean = np.ascontiguousarray(np.asarray(theta_list["mean"], dtype=np.float64))
        out = np.empty(E.shape, order="F")
        _lib.check(self._L.cocons_sim_dense(self._h, _p(T), _p(mean), 1 if classic else 0, E.shape[1], _p(E), _p(out)),
                   "cocons_sim_dense")
        return out

    def sim_cond_core(self, theta_list, locs_pred, x_covariates_pred, locs_unobs, iiderrors, z_col=0):
        lp, Xp, lu = _f(locs_pred), _f(x_covariates_pred), _f(np.asarray(locs_unobs, dtype=np.float64)[:, :2])
        m = Xp.shape[0]
        E = _f(np.asarray(iiderrors, dtype=np.float64).reshape(m, -1))
        T = theta_table(theta_list)
        mean = np.ascontiguousarray(np.asarray(theta_list["mean"], dtype=np.float64))
        out = np.empty(E.shape, order="F")
        _lib.check(self._L.cocons_sim_cond_dense(self._h, _p(T), _p(mean), int(z_col), m, _p(lp), _p(Xp), _p(lu),
                                                 E.shape[1], _p(E), _p(out)), "cocons_sim_cond_dense")
        return out

    def engine_state(self):
        """{"active": last completed operation ran on the engine schedule, "retries": hand-off time-outs so far (each
        repeated once on the plain schedule), "last_abort": code of the last one} -- cocons_fit_engine_state."""
        out = (ctypes.c_int * 3)()
        _lib.check(self._L.cocons_fit_engine_state(self._h, out), "cocons_fit_engine_state")
        return {"active": bool(out[0]), "retries": int(out[1]), "last_abort": int(out[2])}

    def profile_stages(self, theta_list, reps=3):
        """Stage timings (ms) from HIP events on the fit's stream; see cocons_fit_profile."""
        T = theta_table(theta_list)
        mean = np.ascontiguousarray(np.asarray(theta_list["mean"], dtype=np.float64))
        ms = np.zeros(10)
        _lib.check(self._L.cocons_fit_profile(self._h, _p(T), _p(mean), int(reps), _p(ms)), "cocons_fit_profile")
        return {"assembly_ms": ms[0], "cholesky_ms": ms[1], "reduce_ms": ms[2], "eval_ms": ms[3],
                "update_avg_ms": ms[4], "update_launches": int(ms[5]), "update_sum_ms": ms[6],
                "update_flops": ms[7], "dag_ms": ms[8], "dag_flops": ms[9]}


class CoconsTaperFit(CoconsFit):
    """Handle of an optimisation of GetNeg2loglikelihoodTaper: (locs, x_covariates, z, smooth.limits) plus the
    spam pattern of `ref_taper` (colindices / rowpointers, 1-based, symmetric, diagonal stored) and its entries.
    `neg2loglik_core` (inherited) then evaluates the -2 log-likelihood core of the TAPERED covariance through the
    dense factorisation -- spam's value while n^2 doubles fit the device.  Everything else a dense handle
    offers is refused by the library."""

    def __init__(self, locs, x_covariates, z, smooth_limits, colindices, rowpointers, taper_entries, device=-1):
        L = _lib.load()
        self._L = L
        locs, X = _f(locs), _f(x_covariates)
        self.n, self.p = X.shape
        if locs.shape != (self.n, 2):
            raise ValueError("locs must be n x 2")
        z = _f(np.asarray(z, dtype=np.float64).reshape(self.n, -1))
        self.r = z.shape[1]
        self.q = 0
        self.smooth_limits = np.asarray(smooth_limits, dtype=np.float64).copy()
        self.x_covariates = X
        ci = np.ascontiguousarray(np.asarray(colindices, dtype=np.int32))
        rp = np.ascontiguousarray(np.asarray(rowpointers, dtype=np.int32))
        te = np.ascontiguousarray(np.asarray(taper_entries, dtype=np.float64))
        if te.size != ci.size:
            raise ValueError("taper entries and colindices differ in length")
        self._h = L.cocons_fit_create_taper(self.n, self.p, self.r, _p(locs), _p(X), _p(z), _p(self.smooth_limits),
                                            int(device), int(ci.size), _ip(ci), _ip(rp), _p(te))
        if not self._h:
            raise _lib.CoconsHipError("cocons_fit_create_taper failed: " + _lib.last_error())


    def predict_core(self, theta_list, locs_pred, x_covariates_pred, pred_taper, z_col=0):
        """(stochastic, quadform) of the sparse branch of cocoPredict; pred_taper = (colindices, rowpointers,
        entries) of the m x n taper between prediction and observed locations (1-based CSR)."""
        T = theta_table(theta_list)
        mean = np.ascontiguousarray(np.asarray(theta_list["mean"], dtype=np.float64))
        lp, Xp = _f(locs_pred), _f(x_covariates_pred)
        m = Xp.shape[0]
        ci = np.ascontiguousarray(np.asarray(pred_taper[0], dtype=np.int32))
        rp = np.ascontiguousarray(np.asarray(pred_taper[1], dtype=np.int32))
        te = np.ascontiguousarray(np.asarray(pred_taper[2], dtype=np.float64))
        st, qf = np.empty(m), np.empty(m)
        _lib.check(self._L.cocons_predict_taper(self._h, _p(T), _p(mean), int(z_col), m, _p(lp), _p(Xp), int(ci.size),
                                                _ip(ci), _ip(rp), _p(te), _p(st), _p(qf)), "cocons_predict_taper")
        return st, qf


def cocoPredict_sparse(theta_list, locs, newlocs, X_std, X_pred_std, smooth_limits, z, ref_taper, pred_taper,
                       type="pred", fit=None):
    """Sparse branch of cocoPredict, R/predict.R:190-283, from the point where the scaled design matrices, the
    adjusted theta list and the two taper matrices (taper_two = ref_taper, pred_taper; (colindices, rowpointers,
    entries) each) exist."""
    f, own = (fit, False) if fit is not None else (CoconsTaperFit(locs, X_std, z, smooth_limits, *ref_taper), True)
    try:
        st, qf = f.predict_core(theta_list, newlocs, X_pred_std, pred_taper)
    finally:
        if own:
            f.close()
    Xp = np.asarray(X_pred_std, dtype=np.float64)
    systematic = Xp @ np.asarray(theta_list["mean"], dtype=np.float64)                    # :247
    if type == "mean":
        return {"systematic": systematic, "stochastic": st}
    with np.errstate(invalid="ignore"):
        unc = 1 / np.exp(-(Xp @ theta_list["std.dev"])) + np.exp(Xp @ theta_list["nugget"])   # :264-265
    unc = unc - qf                                                                         # :267
    neg = unc < 1e-10
    unc[neg] = np.abs(unc[neg])                                                            # :269-271
    return {"systematic": systematic, "stochastic": st, "sd.pred": np.sqrt(unc)}


def GetNeg2loglikelihoodTaper(theta, par_pos, ref_taper, locs, x_covariates, smooth_limits, z, n, lam, safe=True,
                              fit=None):
    """R/neg2loglikelihood.R:20-53.  `ref_taper` = (colindices, rowpointers, entries) of the spam taper matrix;
    `fit` (optional) a CoconsTaperFit built once from the same data.  cholS has no counterpart: the factorisation
    is dense."""
    tl = getModelLists(theta, par_pos, "diff")
    f, own = (fit, False) if fit is not None else (CoconsTaperFit(locs, x_covariates, z, smooth_limits, *ref_taper), True)
    try:
        try:
            val, _ = f.neg2loglik_core(tl)
        except CholeskyError:
            if safe:
                return 1e6                                  # :35-39
            raise RuntimeError("Cholesky error")
        return val + getPen(n * f.r, lam, tl, smooth_limits)
    finally:
        if own:
            f.close()


def GetNeg2loglikelihoodTaperProfile(theta, par_pos, ref_taper, locs, x_covariates, smooth_limits, z, n, lam,
                                     safe=True, fit=None):
    """R/neg2loglikelihood.R:73-108: std.dev[1] = 0, the marginal variance profiled out."""
    tl = getModelLists(theta, par_pos, "diff")
    sd = np.array(tl["std.dev"], dtype=np.float64, copy=True)
    sd[0] = 0.0                                             # :80
    tl["std.dev"] = sd
    f, own = (fit, False) if fit is not None else (CoconsTaperFit(locs, x_covariates, z, smooth_limits, *ref_taper), True)
    try:
        try:
            _, parts = f.neg2loglik_core(tl)
        except CholeskyError:
            if safe:
                return 1e6
            raise RuntimeError("Cholesky error")
        r = f.r
        logdet, sum_in = parts[0], float(np.sum(parts[1:]))
        return (r * n * np.log(2 * np.pi) + r * n + r * 2 * logdet + r * n * np.log(sum_in / (r * n))
                + getPen(n * r, lam, tl, smooth_limits))    # :102-106
    finally:
        if own:
            f.close()


def _with_fit(fit, locs, x_covariates, z, smooth_limits, x_betas=None):
    if fit is not None:
        return fit, False
    return CoconsFit(locs, x_covariates, z, smooth_limits, x_betas=x_betas), True


def GetNeg2loglikelihood(theta, par_pos, locs, x_covariates, smooth_limits, z, n, lam, safe=True, fit=None):
    """R/neg2loglikelihood.R:183-222.  `fit` (optional) is a CoconsFit built once from the
    same (locs, x_covariates, z, smooth_limits); without it the data are uploaded per call."""
    tl = getModelLists(theta, par_pos, "diff")
    f, own = _with_fit(fit, locs, x_covariates, z, smooth_limits)
    try:
        try:
            val, _ = f.neg2loglik_core(tl)
        except CholeskyError:
            if safe:
                return 1e6                                  # :202-206
            raise RuntimeError("Cholesky error")
        return val + getPen(n * f.r, lam, tl, smooth_limits)
    finally:
        if own:
            f.close()


def GetNeg2loglikelihood_batch(thetas, par_pos, locs, x_covariates, smooth_limits, z, n, lam, safe=True, fit=None):
    """`GetNeg2loglikelihood` at several theta vectors at once -- what optimParallel's workers
    compute in parallel for one gradient (R/optim.R:237-259).  Same values, same `safe` rule."""
    tls = [getModelLists(t, par_pos, "diff") for t in thetas]
    f, own = _with_fit(fit, locs, x_covariates, z, smooth_limits)
    try:
        vals, st = f.neg2loglik_batch_core(tls)
        out = np.empty(len(tls))
        for i, tl in enumerate(tls):
            if st[i] > 0:
                if not safe:
                    raise RuntimeError("Cholesky error")
                out[i] = 1e6
            else:
                out[i] = vals[i] + getPen(n * f.r, lam, tl, smooth_limits)
        return out
    finally:
        if own:
            f.close()


def getHessian_dense(par, par_pos, locs, x_covariates, smooth_limits, z, n, lam, f00=None,
                      eps=np.finfo(float).eps ** 0.25, fit=None):
    """Dense branch of getHessian, R/getFunctions.R:925-1034, from the point where the scaled
    design matrix exists: for every index pair (jj <= ii) three objective values at par shifted by
    eps, H[jj,ii] = 0.5 (f11 - f01 - f10 + f00) / eps^2, then H + t(H) with the diagonal halved.
    The reference farms the 3 P (P+1)/2 evaluations out with parApply (:979); here they form one
    pipelined batch on the GPU."""
    par = np.asarray(par, dtype=np.float64).ravel()
    P = par.size
    f, own = _with_fit(fit, locs, x_covariates, z, smooth_limits)
    try:
        if f00 is None:
            f00 = GetNeg2loglikelihood(par, par_pos, locs, x_covariates, smooth_limits, z, n, lam, fit=f)
        pts, idx = [], []
        for jj in range(P):
            for ii in range(jj, P):
                t01, t10, t11 = par.copy(), par.copy(), par.copy()
                t01[jj] += eps
                t10[ii] += eps
                t11[jj] += eps
                t11[ii] += eps
                pts += [t01, t10, t11]
                idx.append((jj, ii))
        vals = GetNeg2loglikelihood_batch(pts, par_pos, locs, x_covariates, smooth_limits, z, n, lam, fit=f)
        H = np.zeros((P, P))
        for k, (jj, ii) in enumerate(idx):
            f01, f10, f11 = vals[3 * k], vals[3 * k + 1], vals[3 * k + 2]
            H[jj, ii] = 0.5 * ((f11 - f01 - f10 + f00) / (eps * eps))
        H = H + H.T
        H[np.diag_indices(P)] /= 2
        return H
    finally:
        if own:
            f.close()


def GetNeg2loglikelihoodProfile(theta, par_pos, locs, x_covariates, smooth_limits, z, n, x_betas, lam,
                                safe=True, fit=None):
    """R/neg2loglikelihood.R:127-165."""
    tl = getModelLists(theta, par_pos, "diff")
    f, own = _with_fit(fit, locs, x_covariates, z, smooth_limits, x_betas=x_betas)
    try:
        try:
            val, _ = f.neg2loglik_profile_core(tl)
        except CholeskyError:
            if safe:
                return 1e6
            raise RuntimeError("Cholesky error")
        return val + getPen(n * f.r, lam, tl, smooth_limits)
    finally:
        if own:
            f.close()


def GetNeg2loglikelihoodREML(theta, par_pos, locs, x_covariates, x_betas, smooth_limits, z, n, lam,
                             safe=True, fit=None):
    """R/neg2loglikelihood.R:241-291 (x_betas is accepted and, as in the reference, unused)."""
    tl = getModelLists(theta, par_pos, "diff")
    f, own = _with_fit(fit, locs, x_covariates, z, smooth_limits)
    try:
        rank = int(np.linalg.matrix_rank(np.asarray(x_covariates, dtype=np.float64)))   # qr(x)$rank, :270
        try:
            val, _ = f.neg2loglik_reml_core(tl, rank)
        except CholeskyError:
            if safe:
                return 1e6
            raise RuntimeError("Cholesky error")
        return val + getPen((n - rank) * f.r, lam, tl, smooth_limits)
    finally:
        if own:
            f.close()


def getBetas_profile(theta_list, locs, x_covariates, smooth_limits, z, x_betas, fit=None):
    """The "Compute Betas" block of cocoOptim's pml/reml branch, R/optim.R:329-341:
    solve(W, t(V)) %*% rowSums(z) / ncol(z) with V = Sigma^-1 x_betas, W = x_betas' V -- here read
    off the Gram matrix of the bordered factorisation (no second Cholesky, no V)."""
    f, own = _with_fit(fit, locs, x_covariates, z, smooth_limits, x_betas=x_betas)
    try:
        try:
            _, parts = f.neg2loglik_profile_core(theta_list)
        except CholeskyError:
            raise RuntimeError("Cholesky error")
        return parts[2 + f.r: 2 + f.r + f.q].copy()
    finally:
        if own:
            f.close()


def cocoSim_dense(theta_list, locs, X_std, smooth_limits, iiderrors, type="classic", fit=None):
    """Marginal branch of cocoSim for a dense object, R/sim.R:147-172, from the point where the
    scaled design matrix and the theta list exist.  `iiderrors` is the n x nsim matrix of N(0,1)
    draws (the reference draws it with rnorm after set.seed; pass the same numbers for identical
    output).  type = "classic" -> cov_rns_classic, "diff" -> cov_rns.  Returns n x nsim."""
    if type not in ("classic", "diff"):
        raise ValueError("type must be 'classic' or 'diff'")
    E = np.asarray(iiderrors, dtype=np.float64)
    n = np.asarray(X_std).shape[0]
    f, own = _with_fit(fit, locs, X_std, np.zeros(n), smooth_limits)
    try:
        try:
            return f.sim_core(theta_list, E.reshape(n, -1), classic=(type == "classic"))
        except CholeskyError:
            raise RuntimeError("Cholesky error")          # base::chol's error propagates in the reference
    finally:
        if own:
            f.close()


def cocoSim_cond_dense(theta_list, locs, newlocs, newdataset, X_std, X_pred_std, smooth_limits, z, iiderrors,
                       fit=None):
    """Conditional branch of cocoSim (sim.type = "cond") for a dense object, R/sim.R:69-127, from the
    point where the scaled design matrices and the theta list exist.  `newdataset` supplies the
    coordinates of covmat_unobs exactly as the reference passes it (`locs = as.matrix(newdataset)`,
    i.e. its first two columns, :96-99).  Returns m x nsim."""
    f, own = _with_fit(fit, locs, X_std, z, smooth_limits)
    try:
        try:
            return f.sim_cond_core(theta_list, newlocs, X_pred_std, np.asarray(newdataset, dtype=np.float64),
                                   iiderrors)
        except CholeskyError:
            raise RuntimeError("Cholesky error")
    finally:
        if own:
            f.close()


def cocoPredict_dense(theta_list, locs, newlocs, X_std, X_pred_std, smooth_limits, z, type="pred", fit=None):
    """Dense branch of cocoPredict, R/predict.R:136-187, from the point where the scaled
    design matrices and the adjusted theta list exist."""
    f, own = _with_fit(fit, locs, X_std, z, smooth_limits)
    try:
        st, qf = f.predict_core(theta_list, newlocs, X_pred_std)
    finally:
        if own:
            f.close()
    Xp = np.asarray(X_pred_std, dtype=np.float64)
    systematic = Xp @ np.asarray(theta_list["mean"], dtype=np.float64)
    if type == "mean":
        return {"systematic": systematic, "stochastic": st}
    with np.errstate(invalid="ignore"):
        unc = 1 / np.exp(-(Xp @ theta_list["std.dev"])) + np.exp(Xp @ theta_list["nugget"])   # :170-171
    unc = unc - qf                                                                         # :173
    neg = unc < 1e-10
    unc[neg] = np.abs(unc[neg])                                                            # :175-177
    return {"systematic": systematic, "stochastic": st, "sd.pred": np.sqrt(unc)}
