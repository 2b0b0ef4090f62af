"""glue/cocons_hip_glue.c EXECUTED: the `.Call` layer a maintainer drops into the reference's src/ is compiled together
with a small stand-in for R's C API (tests/r_api_stub/r_stub.c -- test infrastructure: typed vectors, attributes, external
pointers with finalizers, the preserve list, MARK_NOT_MUTABLE, Rf_error as a long jump) and linked against
libcocons_hip.so, then driven through ctypes exactly as R's `.Call` would drive it: symbols looked up in the table
R_init_cocons registers, arity checked, errors caught.  R itself is not in this image.

Replaces the reference's generated layer, src/RcppExports.cpp:29-118, and the bodies of R/neg2loglikelihood.R:183-222."""
import ctypes
import math
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB_DIR = os.path.join(ROOT, "tests", "r_api_stub")


def _build():
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    from cocons_amd import _lib
    _lib.load()
    out_dir = os.path.join(STUB_DIR, "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libcocons_glue_stub.so")
    srcs = [os.path.join(ROOT, "glue", "cocons_hip_glue.c"), os.path.join(STUB_DIR, "r_stub.c")]
    deps = srcs + [_lib.LIB_PATH, os.path.join(ROOT, "include", "cocons_hip.h")]
    if os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(d) for d in deps):
        return so
    inc = ["-I", os.path.join(ROOT, "tests", "r_api_decls"), "-I", os.path.join(ROOT, "include")]
    flags = ["-std=gnu11", "-O1", "-g", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-cast-function-type", "-Werror"]
    objs = []
    for src, extra in ((srcs[0], ["-Dgetpid=stub_getpid"]), (srcs[1], [])):
        obj = os.path.join(out_dir, os.path.basename(src) + ".o")
        subprocess.check_call([gcc] + flags + extra + inc + ["-c", src, "-o", obj])
        objs.append(obj)
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call([gcc, "-shared", "-o", so] + objs + ["-L", libdir, "-lcocons_hip", "-Wl,-rpath," + libdir, "-lm"])
    return so


class RStub:
    """ctypes face of the stub: builds R objects from numpy arrays, issues `.Call`s, reads results back."""

    def __init__(self):
        L = ctypes.CDLL(_build(), mode=ctypes.RTLD_GLOBAL)
        vp, ci, cl = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
        for name, res, args in (("stub_nil", vp, []), ("stub_real", vp, [cl]), ("stub_real_matrix", vp, [ci, ci]),
                                ("stub_int", vp, [cl]), ("stub_list", vp, [cl]),
                                ("stub_list_set", None, [vp, cl, vp, ctypes.c_char_p]), ("stub_data", vp, [vp]),
                                ("stub_len", cl, [vp]), ("stub_type", ctypes.c_uint, [vp]), ("stub_nrow", ci, [vp]),
                                ("stub_ncol", ci, [vp]), ("stub_elt", vp, [vp, cl]), ("stub_not_mutable", ci, [vp]),
                                ("stub_preserved", ci, [vp]), ("stub_extptr", vp, [vp]), ("stub_error", ctypes.c_char_p, []),
                                ("stub_protect_depth", ci, []), ("stub_dynamic_symbols", ci, []),
                                ("stub_r_assign_real", vp, [vp, cl, ctypes.c_double]),
                                ("stub_registered_arity", ci, [ctypes.c_char_p]),
                                ("stub_dot_call", vp, [ctypes.c_char_p, ci, ctypes.POINTER(vp)]),
                                ("stub_gc", ci, [ci, ctypes.POINTER(vp)]), ("stub_set_pid", None, [ci]),
                                ("stub_init", None, []), ("R_init_cocons", None, [vp])):
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        self.L = L
        L.stub_init()
        L.R_init_cocons(None)
        self.nil = L.stub_nil()

    # ---- R objects
    def real(self, a):
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 2:
            s = self.L.stub_real_matrix(a.shape[0], a.shape[1])
            flat = np.asfortranarray(a).ravel(order="F")
        else:
            flat = a.ravel()
            s = self.L.stub_real(flat.size)
        if flat.size:
            ctypes.memmove(self.L.stub_data(s), flat.ctypes.data, flat.size * 8)
        return s

    def integer(self, a):
        a = np.ascontiguousarray(np.asarray(a, dtype=np.int32).ravel())
        s = self.L.stub_int(a.size)
        if a.size:
            ctypes.memmove(self.L.stub_data(s), a.ctypes.data, a.size * 4)
        return s

    def named_list(self, d):
        s = self.L.stub_list(len(d))
        for i, (k, v) in enumerate(d.items()):
            self.L.stub_list_set(s, i, v, k.encode() if k is not None else None)
        return s

    def plain_list(self, items):
        s = self.L.stub_list(len(items))
        for i, v in enumerate(items):
            self.L.stub_list_set(s, i, v, None)
        return s

    def theta(self, th, drop_mean=True):
        return self.named_list({k: self.real(v) for k, v in th.items() if not (drop_mean and k == "mean")})

    # ---- .Call
    def call(self, name, *args):
        arr = (ctypes.c_void_p * max(len(args), 1))(*args)
        out = self.L.stub_dot_call(name.encode(), len(args), arr)
        assert self.L.stub_protect_depth() == 0, "protect stack unbalanced after " + name
        if not out:
            raise RuntimeError(self.L.stub_error().decode())
        return out

    def value(self, s):
        t, n = self.L.stub_type(s), self.L.stub_len(s)
        if t == 14:
            a = np.empty(n)
            if n:
                ctypes.memmove(a.ctypes.data, self.L.stub_data(s), n * 8)
            nr, nc = self.L.stub_nrow(s), self.L.stub_ncol(s)
            return a.reshape((nr, nc), order="F") if nr * nc == n and nc > 1 else a
        if t == 13:
            a = np.empty(n, dtype=np.int32)
            if n:
                ctypes.memmove(a.ctypes.data, self.L.stub_data(s), n * 4)
            return a
        if t == 19:
            return [self.value(self.L.stub_elt(s, i)) for i in range(n)]
        if t == 0:
            return None
        return s


@pytest.fixture(scope="module")
def R():
    return RStub()


def test_registration_table_and_host_only_entries(R):
    """R_init_cocons registers the reference's six symbols with the reference's arities (src/RcppExports.cpp:105-113),
    switches dynamic lookup off (:117), and the entries that need no device run: sumsmoothlone against its closed form
    (src/cocons_full.cpp:12-30), argument errors come back as R errors with a message, never as a crash."""
    L = R.L
    for name, arity in {"_cocons_sumsmoothlone": 3, "_cocons_cov_rns": 4, "_cocons_cov_rns_pred": 6,
                        "_cocons_cov_rns_classic": 3, "_cocons_cov_rns_taper_pred": 8, "_cocons_cov_rns_taper": 6}.items():
        assert L.stub_registered_arity(name.encode()) == arity
    assert L.stub_dynamic_symbols() == 0
    x = np.array([0.5, -2.0, 1e-5, 0.0])
    lam, alpha = 2.0, 1e6
    got = R.value(R.call("_cocons_sumsmoothlone", R.real(x), R.real([lam]), R.real([alpha])))[0]
    want = lam * sum(abs(v) if abs(v) > 1e-4 else (math.log1p(math.exp(-alpha * v)) + math.log1p(math.exp(alpha * v))) / alpha
                     for v in x)
    assert abs(got - want) <= 1e-14 * abs(want)
    with pytest.raises(RuntimeError, match="4 arguments, 3 given"):
        R.call("_cocons_cov_rns", R.nil, R.nil, R.nil)
    # theta is looked up BY NAME (src/cocons_full.cpp:47-54): a missing aspect is an R error, not a crash
    th = {"std.dev": R.real(np.zeros(2)), "scale": R.real(np.zeros(2))}
    with pytest.raises(RuntimeError, match="theta has no element 'aniso'"):
        R.call("_cocons_cov_rns", R.named_list(th), R.real(np.zeros((5, 2))), R.real(np.ones((5, 2))), R.real([0.5, 2.5]))
    R.L.R_MakeExternalPtr.restype, R.L.R_MakeExternalPtr.argtypes = ctypes.c_void_p, [ctypes.c_void_p] * 3
    with pytest.raises(RuntimeError, match="handle is NULL"):        # a handle restored from a saved workspace
        R.call("_cocons_hip_engine_state", R.L.R_MakeExternalPtr(None, R.nil, R.nil))


def _problem(g=20, seed=3):
    from cocons_amd import workloads as wl
    locs = wl.grid_locs(g)
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full(scale0=np.log(0.15))
    th["mean"] = np.array([0.2, -0.1, 0.05])
    z = np.random.default_rng(seed).standard_normal(g * g)
    return locs, X, th, z


@pytest.mark.gpu
def test_cov_entries_through_the_glue_vs_oracle(R, oracle):
    """`_cocons_cov_rns`, `_cocons_cov_rns_classic`, `_cocons_cov_rns_pred` as R would call them (R/RcppExports.R:21-46):
    theta as a named list -- with and without its `mean` element, both work in the reference --, results as fresh
    matrices, against the CPU oracle."""
    from cocons_amd import workloads as wl
    locs, X, th, _ = _problem(14)
    sl = R.real(list(wl.SMOOTH_LIMITS))
    So = oracle.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
    for drop in (True, False):
        S = R.value(R.call("_cocons_cov_rns", R.theta(th, drop_mean=drop), R.real(locs), R.real(X), sl))
        assert S.shape == So.shape and np.max(np.abs(S - So) / np.abs(So)) < 2e-12
    Sc = R.value(R.call("_cocons_cov_rns_classic", R.theta(th), R.real(locs), R.real(X)))
    Sco = oracle.cov_rns_classic(th, locs, X)
    assert np.max(np.abs(Sc - Sco) / np.abs(Sco)) < 2e-12
    lp = locs[:37] + 0.013
    lp[5] = locs[9]                                   # a coincident location: the diagonal value (src/cocons_full.cpp:410-414)
    Xp = X[:37] * 0.9
    Xp[5] = X[9]
    C = R.value(R.call("_cocons_cov_rns_pred", R.theta(th), R.real(locs), R.real(lp), R.real(X), R.real(Xp), sl))
    Co = oracle.cov_rns_pred(th, locs, lp, X, Xp, wl.SMOOTH_LIMITS)
    assert C.shape == (37, locs.shape[0]) and np.max(np.abs(C - Co) / np.abs(Co)) < 2e-12


@pytest.mark.gpu
def test_cached_handle_paths_and_status_mapping(R, oracle):
    """`_cocons_hip_fit_cached` + `_cocons_hip_neg2loglik`, the pair behind GetNeg2loglikelihood's unchanged signature
    (R/neg2loglikelihood.R:183-191; glue/R/cocons_hip.R):
      hit            the same R objects again: the same handle, and the keys are preserved and immutable;
      copy / re-key  the same data in other objects (R copied them): the same handle, keys moved to the new objects;
      R-level edit   z[i] <- v on a cached z must duplicate (the stub applies R's rule): new address, other data => a NEW
                     handle and another value -- the stale-handle failure of round 4 (an unsampled element edited in place);
      C-level edit   a vector overwritten in place against R's rules (same address): caught by the verified hit;
      fork           an entry of another pid is dropped, the worker gets its own handle;
      status         a Sigma that is not positive definite comes back as list(status = k > 0, NA), never as an error."""
    from cocons_amd import workloads as wl
    L = R.L
    locs, X, th, z = _problem(20)
    n = z.size
    R.call("_cocons_hip_cache_clear")
    rl, rX, rz, rsl = R.real(locs), R.real(X), R.real(z.reshape(n, 1)), R.real(list(wl.SMOOTH_LIMITS))
    dev = R.integer([0])

    def n2ll(handle, theta):
        st, val = R.value(R.call("_cocons_hip_neg2loglik", handle, R.theta(theta), R.real(theta["mean"])))
        return int(st[0]), float(val[0])

    def want(zz):
        S = oracle.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
        info, ld, quad, _ = oracle.chol_ld(S, (zz - X @ th["mean"]).reshape(-1, 1))
        return n * math.log(2 * math.pi) + 2 * ld + float(quad[0])

    h1 = R.call("_cocons_hip_fit_cached", rl, rX, rz, R.nil, rsl, dev)
    st, v1 = n2ll(h1, th)
    assert st == 0 and abs(v1 - want(z)) <= 1e-8 * abs(v1)
    for s in (rl, rX, rz):
        assert L.stub_not_mutable(s) == 1 and L.stub_preserved(s) == 1
    assert L.stub_preserved(h1) == 1
    # hit
    assert R.call("_cocons_hip_fit_cached", rl, rX, rz, R.nil, rsl, dev) == h1
    # copy of the same data: same handle, keys re-pointed (old objects released, new ones held)
    rz2 = R.real(z.reshape(n, 1))
    assert R.call("_cocons_hip_fit_cached", rl, rX, rz2, R.nil, rsl, dev) == h1
    assert L.stub_preserved(rz) == 0 and L.stub_preserved(rz2) == 1 and L.stub_not_mutable(rz2) == 1
    # R-level modification of ONE element far from any sampled position: R must duplicate an immutable object
    i = 137
    rz3 = L.stub_r_assign_real(rz2, i, z[i] + 0.75)
    assert rz3 != rz2, "a cached key must not be modifiable in place by R code"
    h2 = R.call("_cocons_hip_fit_cached", rl, rX, rz3, R.nil, rsl, dev)
    assert h2 != h1
    z3 = z.copy()
    z3[i] += 0.75
    st, v2 = n2ll(h2, th)
    assert st == 0 and abs(v2 - want(z3)) <= 1e-8 * abs(v2) and abs(v2 - v1) > 1e-6 * abs(v1)
    # the first handle still serves the first data set
    assert R.call("_cocons_hip_fit_cached", rl, rX, rz2, R.nil, rsl, dev) == h1
    # C-level in-place write (same address, against the API's rules): the verified hit refuses the stale handle
    buf = (ctypes.c_double * n).from_address(L.stub_data(rz3))
    buf[4001 % n] += 0.5
    h3 = R.call("_cocons_hip_fit_cached", rl, rX, rz3, R.nil, rsl, dev)
    assert h3 != h2
    z4 = z3.copy()
    z4[4001 % n] += 0.5
    st, v3 = n2ll(h3, th)
    assert st == 0 and abs(v3 - want(z4)) <= 1e-8 * abs(v3)
    # another process (a forked worker): the inherited entries are dropped, the worker creates its own handle
    L.stub_set_pid(os.getpid() + 1)
    try:
        h4 = R.call("_cocons_hip_fit_cached", rl, rX, rz2, R.nil, rsl, dev)
    finally:
        L.stub_set_pid(0)
    assert h4 != h1 and L.stub_preserved(h1) == 0
    # status mapping: not positive definite => status k > 0 and NA, no R error (R/neg2loglikelihood.R:200-206 maps it)
    bad = {k: np.array(v, dtype=float) for k, v in th.items()}
    bad["nugget"][0] = -800.0
    bad["std.dev"][0], bad["scale"][0] = 0.0, 30.0
    st, v = n2ll(h3, bad)
    assert st > 0 and math.isnan(v)
    # a wrong-length mean is an R error with a message
    with pytest.raises(RuntimeError, match="theta\\$mean must have length 3"):
        R.call("_cocons_hip_neg2loglik", h3, R.theta(th), R.real([0.0]))
    # batch entry: list(status = integer(nb), value = double(nb)), one failing point inside
    pts = [th, bad, th]
    res = R.value(R.call("_cocons_hip_neg2loglik_batch", h3, R.plain_list([R.theta(t) for t in pts]),
                         R.plain_list([R.real(t["mean"]) for t in pts])))
    assert res[0][0] == 0 and res[0][1] > 0 and res[0][2] == 0
    assert abs(res[1][0] - v3) <= 1e-12 * abs(v3) and res[1][0] == res[1][2] and math.isnan(res[1][1])
    # dropping the cache releases every key and handle; "collection" then runs the finalizers (cocons_fit_destroy)
    R.call("_cocons_hip_cache_clear")
    for s in (rl, rX, rz2, rz3, h2, h3, h4):
        assert L.stub_preserved(s) == 0
    assert L.stub_gc(0, None) >= 4
    assert L.stub_extptr(h1) is None and L.stub_extptr(h3) is None
