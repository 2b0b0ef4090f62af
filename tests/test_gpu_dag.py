"""The dependency-driven schedule (COCONS_DAG / cocons_debug_tune("dag", 1): ONE persistent launch for every trailing update
and every panel behind the first one, chol.hip dag_kernel) against the classic engine schedule and the CPU oracle, on the
sizes where its special cases live: a last block of one tile (odd number of tiles), front padding with the right-hand
sides in slot rows, right-hand sides under the matrix (n a multiple of 128), two realisations, Profile / REML borders."""
import math
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("COCONS_ENGINE", "1") == "0",
                                 reason="COCONS_ENGINE=0: the dependency-driven schedule needs the diagonal-block engine")]


def _tune(name, value):
    from cocons_amd import _lib
    L = _lib.load()
    _lib.check(L.cocons_debug_tune(name.encode(), int(value)), "cocons_debug_tune")


@pytest.fixture
def dag_on():
    """The DAG schedule for EVERY step (dag_min_tiles = 0: by default only the head of a large factorisation takes it)."""
    _tune("dag", 1)
    _tune("dag_min_tiles", 0)
    yield
    _tune("dag", int(os.environ.get("COCONS_DAG", "1")))
    _tune("dag_min_tiles", int(os.environ.get("COCONS_DAG_MIN_TILES", "2000")))


def _grid(gx, gy):
    from cocons_amd import workloads as wl
    xs, ys = np.linspace(0, 1, gx), np.linspace(0, 1, gy)
    locs = np.array([(x, y) for y in ys for x in xs])
    return locs, wl.design_from_locs(locs)


@pytest.mark.parametrize("gx,gy", [(28, 25), (33, 31), (45, 47), (50, 47), (64, 64), (72, 64)])
def test_dag_vs_classic_and_oracle(oracle, dag_on, gx, gy):
    """n = 700 (6 tiles), 1023 (8 tiles, slots), 2115 (17 tiles: last block of one tile), 2350 (19 tiles), 4096 (32 tiles, no
    padding: right-hand sides in a tile row under the matrix), 4608 (36 tiles): same value as the classic schedule to
    1e-11, as the CPU path to 1e-9 where the oracle finishes in seconds; engine never timed out; DAG really ran."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs, sc = _grid(gx, gy)
    n = locs.shape[0]
    X = sc["std.covs"]
    th = wl.theta_full()
    th["mean"] = np.array([0.1, -0.2, 0.05])
    rng = np.random.default_rng(n)
    z = rng.standard_normal((n, 2)) + (X @ np.array([0.2, 0.3, -0.1]))[:, None]
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
    v_dag, parts_dag = fit.neg2loglik_core(th)
    v_dag2 = fit.neg2loglik_core(th)[0]
    assert v_dag2 == v_dag                                   # same bits: no schedule-dependent order of summation
    st = fit.engine_state()
    assert st["retries"] == 0 and st["active"]
    _tune("dag", 0)
    v_cl, parts_cl = fit.neg2loglik_core(th)
    _tune("dag", 1)
    assert abs(v_dag - v_cl) <= 1e-11 * abs(v_cl), (v_dag, v_cl)
    assert np.allclose(parts_dag, parts_cl, rtol=1e-9, atol=0)
    if n <= 2400:
        S = oracle.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
        info, ld, quad, _ = oracle.chol_ld(S, z - (X @ th["mean"])[:, None])
        want = sum(n * math.log(2 * math.pi) + 2 * ld + float(quad[k]) for k in range(2))
        assert abs(v_dag - want) <= 1e-9 * abs(want)
    fit.close()


def test_dag_profile_reml_and_failure(oracle, dag_on):
    """Profile and REML objectives (q = 3 extra border rows) and a matrix that is not positive definite under the DAG
    schedule, n = 2115."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs, sc = _grid(45, 47)
    n = locs.shape[0]
    X = sc["std.covs"]
    th = wl.theta_full()
    rng = np.random.default_rng(5)
    z = rng.standard_normal(n) + X @ np.array([0.1, 0.4, -0.3])
    pq = wl.par_pos_full()
    tq = wl.theta_vector_from_lists(th, pq)
    lam = (0.1, 0.0, 0.3)
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS, x_betas=X)
    got = ca.GetNeg2loglikelihoodProfile(tq, pq, locs, X, wl.SMOOTH_LIMITS, z, n, X, lam, fit=fit)
    want = oracle.GetNeg2loglikelihoodProfile(tq, pq, locs, X, wl.SMOOTH_LIMITS, z, n, X, lam)
    assert abs(got - want) <= 1e-8 * abs(want)
    got = ca.GetNeg2loglikelihoodREML(tq, pq, locs, X, X, wl.SMOOTH_LIMITS, z, n, lam, fit=fit)
    want = oracle.GetNeg2loglikelihoodREML(tq, pq, locs, X, X, wl.SMOOTH_LIMITS, z, n, lam)
    assert abs(got - want) <= 1e-8 * abs(want)
    assert fit.engine_state()["retries"] == 0
    fit.close()
    th3 = {k: np.zeros(3) for k in th}
    th3["scale"] = np.array([np.log(0.05), 0.0, 0.0])
    th3["nugget"] = np.array([-np.inf, 0.0, 0.0])
    fit3 = ca.CoconsFit(locs, X, z, (1.0, 1.0))
    with pytest.raises(ca.CholeskyError) as ei:
        fit3.neg2loglik_core(th3)
    assert ei.value.minor == 2
    fit3.close()


def test_dag_changing_parameters_stay_reproducible(dag_on):
    """Alternating parameter vectors on one handle (the second buffer and the tile inverses are rewritten every evaluation):
    every value equals the first evaluation of the same parameters bit for bit."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs, sc = _grid(50, 47)
    X = sc["std.covs"]
    z = wl.synthetic_z(locs.shape[0])
    ths = []
    for i in range(4):
        t = wl.theta_full()
        t["scale"][0] += 0.05 * i
        t["std.dev"][1] -= 0.03 * i
        ths.append(t)
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
    first = [fit.neg2loglik_core(t)[0] for t in ths]
    assert len(set(first)) == 4
    rng = np.random.default_rng(0)
    for _ in range(40):
        i = int(rng.integers(4))
        assert fit.neg2loglik_core(ths[i])[0] == first[i]
    assert fit.engine_state()["retries"] == 0
    fit.close()


@pytest.mark.parametrize("g,min_tiles", [(64, 1000), (100, 2000)])
def test_dag_head_then_classic(g, min_tiles):
    """The shipped form: the DAG launch for the head of the factorisation (steps of at least `min_tiles` update tiles), the
    classic schedule behind it -- n = 4096 with a three-step head, and the benchmark size with the default threshold -- against
    the classic schedule throughout; bit-reproducible, engine never timed out."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs = wl.grid_locs(g)
    X = wl.design_from_locs(locs)["std.covs"]
    z = wl.synthetic_z(g * g)
    th = wl.theta_full()
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
    _tune("dag", 0)
    v_cl = fit.neg2loglik_core(th)[0]
    _tune("dag", 1)
    _tune("dag_min_tiles", min_tiles)
    try:
        v = fit.neg2loglik_core(th)[0]
        assert fit.neg2loglik_core(th)[0] == v
        st = fit.profile_stages(th, reps=1)
        assert st["dag_ms"] > 0 and st["dag_flops"] > 0            # the head really ran as one launch
        assert abs(v - v_cl) <= 1e-11 * abs(v_cl), (v, v_cl)
        es = fit.engine_state()
        assert es["retries"] == 0 and es["active"]
    finally:
        _tune("dag", int(os.environ.get("COCONS_DAG", "1")))
        _tune("dag_min_tiles", int(os.environ.get("COCONS_DAG_MIN_TILES", "2000")))
        fit.close()


def test_dag_xcd_quota_changes_nothing_but_who_works():
    """dag_kernel keeps the engine's XCD less than full (COCONS_DAG_XCC_QUOTA workgroups take part there, the others leave
    at once -- room for the driver's save / restore of the queues, chol.hip): which workgroup computes a task does not
    enter the result, so no quota (0), the default and a tiny one (8 of the 255 that land there) give the same bits."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    g = 72
    locs = wl.grid_locs(g)
    X = wl.design_from_locs(locs)["std.covs"]
    z = wl.synthetic_z(g * g)
    th = wl.theta_full()
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
    default = int(os.environ.get("COCONS_DAG_XCC_QUOTA", "208"))
    _tune("dag", 1)
    _tune("dag_min_tiles", 0)
    try:
        vals = []
        for q in (default, 0, 8, default):
            _tune("dag_xcc_quota", q)
            vals.append(fit.neg2loglik_core(th)[0])
            st = fit.profile_stages(th, reps=1)
            assert st["dag_ms"] > 0
        assert len(set(vals)) == 1, vals
        es = fit.engine_state()
        assert es["retries"] == 0 and es["active"]
    finally:
        _tune("dag_xcc_quota", default)
        _tune("dag", int(os.environ.get("COCONS_DAG", "1")))
        _tune("dag_min_tiles", int(os.environ.get("COCONS_DAG_MIN_TILES", "2000")))
        fit.close()


@pytest.mark.parametrize("gx,gy,min_tiles", [(45, 47, 0), (72, 64, 0), (100, 100, 2000)])
def test_dag_xcd_aware_order_same_bits(gx, gy, min_tiles):
    """Round 6: the XCD-aware task order of the persistent launch (list positions dealt to the XCDs in chunks, the far tiles of a
    step dealt so that one XCD's tiles in flight form a compact block, the same number of workgroups from every XCD) decides
    WHO computes a tile and WHEN, never what is summed in which order: the value and its parts are bit-identical to the
    one-counter, column-major order of rounds 4-5, for either far-tile order under either deal and for other block shapes;
    no hand-off times out."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs, sc = _grid(gx, gy)
    n = locs.shape[0]
    X = sc["std.covs"]
    th = wl.theta_full()
    th["mean"] = np.array([0.1, -0.2, 0.05])
    rng = np.random.default_rng(n)
    z = rng.standard_normal((n, 2)) + (X @ np.array([0.2, 0.3, -0.1]))[:, None]
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
    try:
        _tune("dag", 1)
        _tune("dag_min_tiles", min_tiles)
        _tune("dag_xcd", 0)
        _tune("dag_order", 0)
        v0, p0 = fit.neg2loglik_core(th)
        assert fit.engine_state()["active"]
        for xcd, order, bw, bh in ((0, 1, 16, 16), (1, 0, 16, 16), (1, 1, 16, 16), (1, 1, 8, 24), (1, 1, 5, 7)):
            _tune("dag_xcd", xcd)
            _tune("dag_order", order)
            _tune("dag_bw", bw)
            _tune("dag_bh", bh)
            for _ in range(2):
                v, p = fit.neg2loglik_core(th)
                assert v == v0 and np.array_equal(p, p0), (xcd, order, bw, bh)
        st = fit.engine_state()
        assert st["retries"] == 0 and st["active"], st
    finally:
        _tune("dag_xcd", int(os.environ.get("COCONS_DAG_XCD", "1")))
        _tune("dag_order", int(os.environ.get("COCONS_DAG_ORDER", "1")))
        _tune("dag_bw", int(os.environ.get("COCONS_DAG_BW", "16")))
        _tune("dag_bh", int(os.environ.get("COCONS_DAG_BH", "16")))
        _tune("dag_min_tiles", int(os.environ.get("COCONS_DAG_MIN_TILES", "2000")))
        fit.close()
