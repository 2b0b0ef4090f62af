"""GPU tests on the BASELINE configurations at (or near) full size.  Parity against the CPU
oracle where the oracle finishes in seconds (C1 n=400, C2/C4 n=4096); size-independent
properties at the sizes the oracle cannot reach quickly (C3 n=10 000, C5 n=m=8192)."""
import math
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _grid_problem(g):
    from cocons_amd import workloads as wl
    locs = wl.grid_locs(g)
    X = wl.design_from_locs(locs)["std.covs"]
    return locs, X, wl.theta_full(), wl.synthetic_z(g * g)


def test_c1_holes_plumbing(oracle, golden_dir):
    """C1: 400 `holes` rows, stationary Matern nu=1.5 through cov_rns_classic and cov_rns,
    one -2 loglik call; GPU vs CPU."""
    import cocons_amd as ca
    d = np.loadtxt(os.path.join(golden_dir, "holes_train400.csv"), delimiter=",", skiprows=1)
    n = d.shape[0]
    locs, z = d[:, :2], d[:, 4]
    X = np.ones((n, 1))
    sc = math.log(0.2)
    th = {"mean": np.zeros(1), "std.dev": np.zeros(1), "scale": np.array([sc]), "aniso": np.zeros(1),
          "tilt": np.zeros(1), "smooth": np.array([math.log(1.5)]), "nugget": np.array([math.log(0.01)])}
    Sc = ca.cov_rns_classic(th, locs, X)
    assert np.max(np.abs(Sc - oracle.cov_rns_classic(th, locs, X)) / np.abs(Sc)) < 2e-12
    th["smooth"] = np.zeros(1)
    S = ca.cov_rns(th, locs, X, (1.5, 1.5))
    assert np.max(np.abs(S - Sc) / np.abs(S)) < 1e-12            # Bessel branch == closed form
    pp = {"mean": 0.0, "std.dev": [True], "scale": [True], "aniso": 0.0, "tilt": 0.0, "smooth": 0.0,
          "nugget": [True]}
    tv = np.array([0.0 + sc, 0.0 - sc, math.log(0.01)])
    got = ca.GetNeg2loglikelihood(tv, pp, locs, X, (1.5, 1.5), z, n, (0, 0, 0))
    want = oracle.GetNeg2loglikelihood(tv, pp, locs, X, (1.5, 1.5), z, n, (0, 0, 0))
    assert abs(got - want) <= 1e-8 * abs(want)


def test_c2_grid4096_vs_cpu(oracle):
    """C2: 64x64 grid, full nonstationary model, dense -2 loglik: entrywise Sigma and the value."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs, X, th, z = _grid_problem(64)
    n = 4096
    S = ca.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
    So = oracle.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
    big = np.abs(So) > 1e-280
    assert np.max(np.abs(S[big] - So[big]) / np.abs(So[big])) < 2e-12
    pp = wl.par_pos_full()
    tv = wl.theta_vector_from_lists(th, pp)
    got = ca.GetNeg2loglikelihood(tv, pp, locs, X, wl.SMOOTH_LIMITS, z, n, (0.1, 0.1, 0.1))
    want = oracle.GetNeg2loglikelihood(tv, pp, locs, X, wl.SMOOTH_LIMITS, z, n, (0.1, 0.1, 0.1))
    assert abs(got - want) <= 1e-8 * abs(want)
    # stress set: no nugget, shorter range (worse conditioning); tolerance still the north star's
    th2 = wl.theta_full(nugget=False, scale0=np.log(0.02))
    tv2 = wl.theta_vector_from_lists(th2, pp)
    pp2 = wl.par_pos_full()
    pp2["nugget"] = -np.inf
    tv2 = tv2[:-1]
    got2 = ca.GetNeg2loglikelihood(tv2, pp2, locs, X, wl.SMOOTH_LIMITS, z, n, (0, 0, 0))
    want2 = oracle.GetNeg2loglikelihood(tv2, pp2, locs, X, wl.SMOOTH_LIMITS, z, n, (0, 0, 0))
    assert abs(got2 - want2) <= 1e-8 * abs(want2)


def test_c3_n10000_properties():
    """C3 at full size, properties the domain offers (the CPU value at n=10 000 is checked by
    bench.py's cpu_baseline leg, `parity_rel_err_vs_cpu`):
      - permuting the locations leaves -2 loglik unchanged,
      - scaling z by c scales every quadratic form by c^2 and leaves log det unchanged,
      - shifting the std.dev and nugget intercepts by 2 log s scales Sigma by s^2:
        sum(log diag chol) moves by n log s, the quadratic form by 1/s^2."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs, X, th, z = _grid_problem(100)
    n = 10000
    fit = ca.CoconsFit(locs, X, np.column_stack([z, 3.0 * z]), wl.SMOOTH_LIMITS)
    val, parts = fit.neg2loglik_core(th)
    assert np.isfinite(val)
    assert abs(parts[2] - 9.0 * parts[1]) < 1e-10 * parts[2]
    perm = np.random.default_rng(1).permutation(n)
    fitp = ca.CoconsFit(locs[perm], X[perm], np.column_stack([z, 3.0 * z])[perm], wl.SMOOTH_LIMITS)
    valp, partsp = fitp.neg2loglik_core(th)
    assert abs(partsp[0] - parts[0]) < 1e-10 * abs(parts[0])
    assert abs(partsp[1] - parts[1]) < 1e-9 * abs(parts[1])
    assert abs(valp - val) < 1e-9 * abs(val)
    s = 1.7
    th2 = {k: np.array(v, dtype=float) for k, v in th.items()}
    th2["std.dev"][0] += 2 * math.log(s)
    th2["nugget"][0] += 2 * math.log(s)
    val2, parts2 = fit.neg2loglik_core(th2)
    assert abs(parts2[0] - (parts[0] + n * math.log(s))) < 1e-10 * abs(parts2[0])
    assert abs(parts2[1] - parts[1] / s ** 2) < 1e-9 * abs(parts2[1])


def test_c3_n10000_vs_cpu(oracle):
    """C3 at full size against the CPU oracle: cov_rns restatement (serial, ~7 s) + LAPACK dpotrf / dtrtrs on every host
    core, the same inputs, -2 loglik within the north star's 1e-8 relative; log det and the quadratic form separately
    (R/neg2loglikelihood.R:183-222)."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    from scipy.linalg import lapack, solve_triangular
    locs, X, th, z = _grid_problem(100)
    n = 10000
    th = {k: np.array(v, dtype=float) for k, v in th.items()}
    th["mean"] = np.array([0.2, -0.1, 0.05])
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
    val, parts = fit.neg2loglik_core(th)
    assert fit.engine_state()["active"] or os.environ.get("COCONS_ENGINE", "1") == "0"   # the shipped schedule (engine + DAG head), not a fall-back
    S = oracle.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
    R, info = lapack.dpotrf(S, lower=0, clean=0, overwrite_a=1)
    assert info == 0
    logdet = float(np.sum(np.log(np.diag(R))))
    y = solve_triangular(R, z - X @ th["mean"], trans="T", lower=False, check_finite=False)
    quad = float(y @ y)
    want = n * math.log(2 * math.pi) + 2 * logdet + quad
    assert abs(parts[0] - logdet) <= 1e-10 * abs(logdet)
    assert abs(parts[1] - quad) <= 1e-8 * abs(quad)
    assert abs(val - want) <= 1e-8 * abs(want)
    # the public closure with the penalty on top (host arithmetic) against the oracle's, same Sigma
    pp = wl.par_pos_full()
    pp["mean"] = [True, True, True]
    tv = np.concatenate([th["mean"], wl.theta_vector_from_lists(th, wl.par_pos_full())])
    lam = (0.1, 0.05, 0.2)
    got = ca.GetNeg2loglikelihood(tv, pp, locs, X, wl.SMOOTH_LIMITS, z, n, lam, fit=fit)
    tl = oracle.getModelLists(tv, pp, "diff")
    assert abs(got - (want + oracle.getPen(n, lam, tl, wl.SMOOTH_LIMITS))) <= 1e-8 * abs(want)


def test_c4_optimizer_in_the_loop(oracle):
    """C4 pattern at a size the oracle affords: L-BFGS-B with central differences
    (ndeps = eps^(1/4), 1 + 2P evaluations per gradient) on the GPU objective; the GPU and the
    CPU objective agree at the start and at the end point, and the optimiser decreased -2 loglik."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from optim_loop import lbfgsb_central
    locs, X, th, _ = _grid_problem(24)
    n = 576
    rng = np.random.default_rng(3)
    S = oracle.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
    z = np.linalg.cholesky(S) @ rng.standard_normal(n)
    pp = wl.par_pos_full()
    t0 = wl.theta_vector_from_lists(th, pp) + 0.1
    lam = (0.0, 0.0, 0.0)
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)

    def f_gpu(t):
        return ca.GetNeg2loglikelihood(t, pp, locs, X, wl.SMOOTH_LIMITS, z, n, lam, fit=fit)

    res = lbfgsb_central(f_gpu, t0, lower=t0 - 3, upper=t0 + 3, max_evals=70)
    assert res["nfev"] >= 34 and res["fun"] < f_gpu(t0)
    for t in (t0, res["x"]):
        a, b = f_gpu(t), oracle.GetNeg2loglikelihood(t, pp, locs, X, wl.SMOOTH_LIMITS, z, n, lam)
        assert abs(a - b) <= 1e-8 * abs(b)


def test_c4_optimizer_in_the_loop_n4096(oracle):
    """C4 as BASELINE states it: 64x64 grid (n = 4096), full nonstationary model, P = 16 free
    parameters, L-BFGS-B with central differences, ~50 objective evaluations on the GPU (the 2P
    gradient points through the batch entry).  GPU objective vs the CPU oracle at the start and at
    the end point (R/optim.R:237-259)."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from optim_loop import lbfgsb_central
    locs, X, th, z = _grid_problem(64)
    n = 4096
    pp = wl.par_pos_full()
    t0 = wl.theta_vector_from_lists(th, pp) + 0.1
    lam = (0.0, 0.0, 0.0)
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)

    def f_gpu(t):
        return ca.GetNeg2loglikelihood(t, pp, locs, X, wl.SMOOTH_LIMITS, z, n, lam, fit=fit)

    def f_batch(ts):
        return ca.GetNeg2loglikelihood_batch(ts, pp, locs, X, wl.SMOOTH_LIMITS, z, n, lam, fit=fit)

    res = lbfgsb_central(f_gpu, t0, lower=t0 - 3, upper=t0 + 3, max_evals=50, fn_batch=f_batch)
    assert res["nfev"] >= 34 and res["fun"] < f_gpu(t0)
    for t in (t0, res["x"]):
        a, b = f_gpu(t), oracle.GetNeg2loglikelihood(t, pp, locs, X, wl.SMOOTH_LIMITS, z, n, lam)
        assert abs(a - b) <= 1e-8 * abs(b)


def test_getHessian_batch_vs_cpu(oracle):
    """getHessian: 3 P (P+1)/2 objective evaluations as one GPU batch against
    the serial CPU restatement.  The finite-difference quotient divides differences of O(1e3)
    numbers by eps^2 = 1.5e-8, so objective parity of 1e-13 relative shows up as ~1e-2 absolute
    in H: compare relative to the largest entry."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs, X, th, _ = _grid_problem(12)
    n = 144
    rng = np.random.default_rng(8)
    S = oracle.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
    z = np.linalg.cholesky(S) @ rng.standard_normal(n)
    pp = wl.par_pos_full()
    pp["tilt"] = 0.0
    pp["aniso"] = 0.0
    t0 = np.concatenate([wl.theta_vector_from_lists(th, wl.par_pos_full())[:6],
                         wl.theta_vector_from_lists(th, wl.par_pos_full())[12:]])
    lam = (0.0, 0.0, 0.0)
    Hg = ca.getHessian_dense(t0, pp, locs, X, wl.SMOOTH_LIMITS, z, n, lam)
    Hc = oracle.getHessian_dense(t0, pp, locs, X, wl.SMOOTH_LIMITS, z, n, lam)
    assert Hg.shape == (10, 10) and np.allclose(Hg, Hg.T)
    assert np.max(np.abs(Hg - Hc)) < 1e-3 * np.max(np.abs(Hc))


def test_c5_predict_8192_properties():
    """C5 at full size (n = m = 8192).  Property: predicting AT the training locations, the
    cross-covariance rows are rows of Sigma (coincident points take the diagonal value,
    src/cocons_full.cpp:410-414), so  C Sigma^-1 r = r  and  diag(C Sigma^-1 C') = diag(Sigma)."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs = wl.grid_locs(128, 64)
    sc = wl.design_from_locs(locs)
    X = sc["std.covs"]
    th = wl.theta_full()
    th["mean"] = np.array([0.3, -0.1, 0.2])
    z = wl.synthetic_z(8192)
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
    st, qf = fit.predict_core(th, locs, X)
    resid = z - X @ th["mean"]
    assert np.max(np.abs(st - resid)) < 1e-8 * np.max(np.abs(resid))
    diag = 1 / np.exp(-(X @ th["std.dev"])) + np.exp(X @ th["nugget"])
    assert np.max(np.abs(qf - diag)) < 1e-8 * np.max(diag)
    # and the shifted prediction grid of the benchmark config runs to finite, sane values
    lp = locs + np.array([0.5 / 127, 0.5 / 63])
    Xp = wl.design_from_locs(lp, sc["mean.vector"], sc["sd.vector"])["std.covs"]
    out = ca.cocoPredict_dense(th, locs, lp, X, Xp, wl.SMOOTH_LIMITS, z, fit=fit)
    assert np.all(np.isfinite(out["stochastic"])) and np.all(np.isfinite(out["sd.pred"]))
    dvar = 1 / np.exp(-(Xp @ th["std.dev"])) + np.exp(Xp @ th["nugget"])
    assert np.all(out["sd.pred"] ** 2 <= dvar * (1 + 1e-9))


def test_c5_predict_8192_vs_cpu(oracle):
    """C5 at full size against the CPU oracle: the GPU predicts at all m = 8192 shifted-grid locations; 512 of them
    (every 16th) are predicted by oracle.cocoPredict_dense -- cov_rns + cov_rns_pred restatements and LAPACK's LU solve,
    R/predict.R:136-183 -- on the same training set.  Every prediction row's solve is independent of the others, so the
    rows must agree one by one: stochastic part and prediction standard deviation."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    locs, X, th, z, lp, Xp = _c5_problem()
    got = ca.cocoPredict_dense(th, locs, lp, X, Xp, wl.SMOOTH_LIMITS, z)
    idx = np.arange(0, 8192, 16)
    assert idx.size == 512
    want = oracle.cocoPredict_dense(th, locs, lp[idx], X, Xp[idx], wl.SMOOTH_LIMITS, z)
    scale = np.max(np.abs(want["stochastic"]))
    assert np.max(np.abs(got["stochastic"][idx] - want["stochastic"])) <= 1e-8 * scale
    assert np.max(np.abs(got["systematic"][idx] - want["systematic"])) <= 1e-13 * max(1.0, np.max(np.abs(want["systematic"])))
    assert np.max(np.abs(got["sd.pred"][idx] - want["sd.pred"]) / want["sd.pred"]) <= 1e-8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_worker(rank, world, port, g, out_dir, predict_first=False, group=None):
    sys.path.insert(0, ROOT)
    if group is not None:
        os.environ["COCONS_SHARD_GROUP"] = str(group)      # read once, when the library is first used
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch                      # noqa: F401
    import torch.distributed as dist
    from cocons_amd import workloads as wl
    from cocons_amd.shard import ShardedFit
    dist.init_process_group("gloo", rank=rank, world_size=world)
    locs = wl.grid_locs(g)
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full()
    th["mean"] = np.array([0.1, -0.2, 0.05])
    z = np.column_stack([wl.synthetic_z(g * g), wl.synthetic_z(g * g, seed=5)])
    fit = ShardedFit(locs, X, z, wl.SMOOTH_LIMITS, device=0)     # every rank on the one GPU of the box
    if predict_first:
        # grows the handle's border (leading dimension) before the sharded evaluation: the packed panels
        # must still be sized by the rows in use (round-1 advisor finding)
        fit.predict_core(th, locs[:300] + 1e-3, X[:300])
    fit.init_host_transport(dist, rank, world)
    assert fit.world() == world
    val, parts = fit.neg2loglik_core(th)          # the library's own schedule; gloo is only the wire
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.concatenate([[val], parts]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.shared_gpu
@pytest.mark.parametrize("world,g,predict_first,group", [(2, 40, False, None), (3, 50, False, None), (4, 15, False, None),
                                                         (2, 40, True, None), (2, 100, False, None),
                                                         (2, 40, False, 1), (3, 50, False, 2), (4, 50, False, 3)])
def test_native_sharded_evaluation_shared_gpu(tmp_path, world, g, predict_first, group):
    """The production sharded path -- the schedule inside the HIP library (sharded_eval: panel ownership,
    look-ahead, communication stream, double-buffered exchange, final all-reduce) -- with `world` ranks
    sharing this box's single GPU.  RCCL refuses several ranks on one device, so the library's
    broadcast / all-reduce hooks are served by gloo through host memory; everything else is the code the
    RCCL build runs.  Must reproduce the single-GPU value.  (2, 100) is BASELINE config C3 at its full
    size n = 10 000 in sharded form.  `group` = panels per ownership group (None: the library's default, 4;
    1: the cyclic deal of rounds 1-2)."""
    import torch.multiprocessing as mp
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    mp.spawn(_shard_worker, args=(world, _free_port(), g, str(tmp_path), predict_first, group), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), "rank%d.npy" % r)) for r in range(world)]
    for r in res[1:]:
        assert np.array_equal(r, res[0])
    locs = wl.grid_locs(g)
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full()
    th["mean"] = np.array([0.1, -0.2, 0.05])
    z = np.column_stack([wl.synthetic_z(g * g), wl.synthetic_z(g * g, seed=5)])
    val, parts = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS).neg2loglik_core(th)
    assert abs(res[0][0] - val) < 1e-10 * abs(val)
    assert np.allclose(res[0][1:], parts, rtol=1e-10, atol=0)


def test_native_sharded_rccl_one_rank_and_multi_handle():
    """RCCL itself, as far as one GPU allows: a communicator of ONE rank created from a unique id
    (cocons_comm_unique_id / cocons_fit_comm_init) and the one-process multi-GPU handle over the device
    list [0] (cocons_multi_create -> ncclCommInitAll).  Both run the sharded schedule with ncclBroadcast /
    ncclAllReduce on the library's communication stream and must reproduce the plain value."""
    import ctypes
    import cocons_amd as ca
    from cocons_amd import _lib, workloads as wl
    from cocons_amd.shard import MultiFit, ShardedFit
    g = 36
    locs = wl.grid_locs(g)
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full()
    th["mean"] = np.array([0.1, -0.2, 0.05])
    z = np.column_stack([wl.synthetic_z(g * g), wl.synthetic_z(g * g, seed=5)])
    want, wparts = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS).neg2loglik_core(th)
    fit = ShardedFit(locs, X, z, wl.SMOOTH_LIMITS, device=0)
    L = _lib.load()
    idb = (ctypes.c_ubyte * _lib.UNIQUE_ID_BYTES)()
    _lib.check(L.cocons_comm_unique_id(ctypes.cast(idb, ctypes.c_void_p)), "cocons_comm_unique_id")
    _lib.check(L.cocons_fit_comm_init(fit._h, 1, 0, ctypes.cast(idb, ctypes.c_void_p)), "cocons_fit_comm_init")
    for _ in range(2):
        got, parts = fit.neg2loglik_core(th)
        assert abs(got - want) < 1e-10 * abs(want) and np.allclose(parts, wparts, rtol=1e-10, atol=0)
    fit.close()
    mf = MultiFit(locs, X, z, wl.SMOOTH_LIMITS, devices=[0])
    got, parts = mf.neg2loglik_core(th)
    assert abs(got - want) < 1e-10 * abs(want) and np.allclose(parts, wparts, rtol=1e-10, atol=0)
    # a theta whose Sigma is not positive definite comes back as the failing minor on this path too
    bad = {k: np.array(v, dtype=float) for k, v in th.items()}
    bad["nugget"][0] = -800.0
    bad["std.dev"][0], bad["scale"][0] = 0.0, 30.0
    with pytest.raises(_lib.CholeskyError):
        mf.neg2loglik_core(bad)
    mf.close()


def _rccl_worker(rank, world, port, g, out_dir, comm2):
    sys.path.insert(0, ROOT)
    os.environ["COCONS_SHARD_COMM2"] = str(comm2)             # read once, when the library is first used
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch                      # noqa: F401
    import torch.distributed as dist
    from cocons_amd import workloads as wl
    from cocons_amd.shard import ShardedFit
    dist.init_process_group("gloo", rank=rank, world_size=world)      # (moves the 128-byte unique id, nothing else)
    locs = wl.grid_locs(g)
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full()
    th["mean"] = np.array([0.1, -0.2, 0.05])
    z = np.column_stack([wl.synthetic_z(g * g), wl.synthetic_z(g * g, seed=5)])
    fit = ShardedFit(locs, X, z, wl.SMOOTH_LIMITS, device=rank)       # one rank per GPU
    fit.init_rccl(dist, rank, world)
    info = fit.comm_info()
    assert info["count"] == world and info["rank"] == rank and info["device"] == rank, info
    for _ in range(3):
        val, parts = fit.neg2loglik_core(th)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.concatenate([[val], parts]))
    dist.barrier()
    fit.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("comm2", [0, 1])
def test_native_sharded_rccl_two_gpus(tmp_path, comm2):
    """The sharded evaluation over RCCL with MORE THAN ONE RANK -- one process per GPU, ncclCommInitRank from a shared unique
    id, broadcast / all-gather / all-reduce over xGMI -- which a one-GPU box cannot run: skipped there, and the first thing
    to run on a node with two devices (the advisor's round-5 finding: the collective order of round 5 and the split
    communicator, COCONS_SHARD_COMM2=1, have never met a second rank).  Both forms must reproduce the single-GPU value on
    every rank; then the one-process handle over the device list [0, 1] (cocons_multi_neg2loglik_dense, fit_comm_init's twin)."""
    import torch.multiprocessing as mp
    import cocons_amd as ca
    from cocons_amd import _lib, workloads as wl
    if _lib.load().cocons_device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    g, world = 50, 2
    mp.spawn(_rccl_worker, args=(world, _free_port(), g, str(tmp_path), comm2), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), "rank%d.npy" % r)) for r in range(world)]
    assert np.array_equal(res[0], res[1])
    locs = wl.grid_locs(g)
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full()
    th["mean"] = np.array([0.1, -0.2, 0.05])
    z = np.column_stack([wl.synthetic_z(g * g), wl.synthetic_z(g * g, seed=5)])
    val, parts = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS).neg2loglik_core(th)
    assert abs(res[0][0] - val) < 1e-10 * abs(val)
    assert np.allclose(res[0][1:], parts, rtol=1e-10, atol=0)
    if comm2 == 0:
        from cocons_amd.shard import MultiFit
        mf = MultiFit(locs, X, z, wl.SMOOTH_LIMITS, devices=[0, 1])
        got, mparts = mf.neg2loglik_core(th)
        assert abs(got - val) < 1e-10 * abs(val) and np.allclose(mparts, parts, rtol=1e-10, atol=0)
        mf.close()


def _predict_shard_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch                      # noqa: F401
    import torch.distributed as dist
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    from cocons_amd.shard import sharded_predict_core
    dist.init_process_group("gloo", rank=rank, world_size=world)
    locs, X, th, z, lp, Xp = _c5_problem()
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS, device=0)
    st, qf = sharded_predict_core(fit, th, lp, Xp, dist, rank, world)
    np.save(os.path.join(out_dir, "pred%d.npy" % rank), np.stack([st, qf]))
    dist.barrier()
    dist.destroy_process_group()


def _c5_problem():
    from cocons_amd import workloads as wl
    locs = wl.grid_locs(128, 64)
    sc = wl.design_from_locs(locs)
    X = sc["std.covs"]
    th = wl.theta_full()
    th["mean"] = np.array([0.3, -0.1, 0.2])
    z = wl.synthetic_z(8192)
    lp = locs + np.array([0.5 / 127, 0.5 / 63])
    Xp = wl.design_from_locs(lp, sc["mean.vector"], sc["sd.vector"])["std.covs"]
    return locs, X, th, z, lp, Xp


@pytest.mark.shared_gpu
def test_c5_predict_split_two_ranks(tmp_path):
    """C5 as BASELINE states it: n_train = m_pred = 8192 with the prediction locations split over
    2 ranks (cocons_amd.shard.sharded_predict_core with real fit handles; both ranks share this
    box's GPU, gloo carries the final gather).  Must equal the unsharded call: each prediction
    row's solve is independent of the other rows (R/predict.R:150-173)."""
    import torch.multiprocessing as mp
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    world = 2
    mp.spawn(_predict_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), "pred%d.npy" % r)) for r in range(world)]
    assert np.array_equal(res[0], res[1])
    locs, X, th, z, lp, Xp = _c5_problem()
    st, qf = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS).predict_core(th, lp, Xp)
    assert np.max(np.abs(res[0][0] - st)) <= 1e-10 * np.max(np.abs(st))
    assert np.max(np.abs(res[0][1] - qf)) <= 1e-10 * np.max(np.abs(qf))


def test_c5_predict_split_native_multi_handle():
    """C5 through the one-process multi-device handle (cocons_multi_predict_dense): the device list [0, 0]
    splits the 8192 prediction locations over two fit handles driven by two host threads (no communicator is
    needed for this entry point); must equal the unsharded call."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    from cocons_amd.shard import MultiFit
    locs, X, th, z, lp, Xp = _c5_problem()
    mf = MultiFit(locs, X, z, wl.SMOOTH_LIMITS, devices=[0, 0])
    st, qf = mf.predict_core(th, lp, Xp)
    with pytest.raises(ca.CoconsHipError):
        mf.neg2loglik_core(th)                 # no RCCL communicator over a repeated device
    mf.close()
    st1, qf1 = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS).predict_core(th, lp, Xp)
    assert np.max(np.abs(st - st1)) <= 1e-10 * np.max(np.abs(st1))
    assert np.max(np.abs(qf - qf1)) <= 1e-10 * np.max(np.abs(qf1))


def _optim_worker(rank, world, port, g, nevals, out_dir):
    """One of cocoOptim's worker processes (R/optim.R:117-121, :234-259): its own library instance and its own fit
    handle on device 0, evaluating the objective at its share of the parameter points while the other workers do the
    same on the same GPU."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import time
    import torch                      # noqa: F401
    import torch.distributed as dist
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    dist.init_process_group("gloo", rank=rank, world_size=world)
    locs, X, th, z = _grid_problem(g)
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS, device=0)
    fit.neg2loglik_core(th)                         # first call: allocations, code load
    dist.barrier()                                  # every worker starts its evaluations at the same time
    t0 = time.perf_counter()
    vals = []
    for e in range(nevals):
        t = {k: np.array(v, dtype=float).copy() for k, v in th.items()}
        t["std.dev"][1] += 0.01 * (rank * nevals + e)          # every point of every worker is a different theta
        vals.append(fit.neg2loglik_core(t)[0])
    wall = time.perf_counter() - t0
    st = fit.engine_state()
    np.save(os.path.join(out_dir, "worker%d.npy" % rank),
            np.array(vals + [st["retries"], st["last_abort"], 1.0 if st["active"] else 0.0, wall]))
    dist.barrier()
    fit.close()
    dist.destroy_process_group()


def test_multi_handle_replica_batch(oracle):
    """Replica mode for one R process (R/optim.R:256-259, R/getFunctions.R:979-1016): the points of a
    finite-difference gradient dealt over the devices of a multi handle -- here the device list [0, 0], two fits with
    their own slots on the one GPU, driven by two host threads -- against single evaluations and the CPU oracle."""
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    from cocons_amd.shard import MultiFit
    locs, X, th, z = _grid_problem(40)
    pts = []
    for i in range(9):
        t = {k: np.array(v, dtype=float).copy() for k, v in th.items()}
        t["scale"][1] += 0.02 * i
        t["mean"] = np.array([0.1 * i, 0.0, -0.05])
        pts.append(t)
    bad = {k: np.array(v, dtype=float).copy() for k, v in th.items()}
    bad["nugget"][0] = -800.0
    bad["std.dev"][0], bad["scale"][0] = 0.0, 30.0
    pts.insert(4, bad)                                        # a point whose Sigma is not positive definite
    mf = MultiFit(locs, X, z, wl.SMOOTH_LIMITS, devices=[0, 0])
    assert mf.comm_ranks() == (2, 0)                          # no communicator over a repeated device, none needed
    vals, st = mf.neg2loglik_batch_core(pts)
    mf.close()
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
    for i, t in enumerate(pts):
        if i == 4:
            assert st[i] > 0 and np.isnan(vals[i])
            continue
        assert st[i] == 0
        one = fit.neg2loglik_core(t)[0]                       # (engine schedule; the batch slots run the plain one)
        assert abs(vals[i] - one) <= 1e-12 * abs(one)
        if i in (0, 7):                                       # the CPU restatement of the same core (no penalty)
            S = oracle.cov_rns(t, locs, X, wl.SMOOTH_LIMITS)
            info, ld, quad, _ = oracle.chol_ld(S, (z - X @ t["mean"]).reshape(-1, 1))
            want = z.size * math.log(2 * math.pi) + 2 * ld + float(quad[0])
            assert abs(vals[i] - want) <= 1e-8 * abs(want)
    fit.close()


@pytest.mark.shared_gpu
@pytest.mark.parametrize("g,nevals", [(64, 10), (100, 6)])
def test_worker_processes_share_one_gpu(tmp_path, record_property, g, nevals):
    """The reference's own calling pattern on the HIP path: `ncores` worker PROCESSES (here 3), each with a handle of
    its own on the one device, evaluating concurrently at n = 4096 (R/optim.R:117-121, 234-259).  Every value must
    equal the single-process value; each process's engine needs a CU to itself while the others' updates fill the
    chip, so hand-off time-outs are allowed here -- each costs one repeat on the plain schedule and is counted; the
    back-off bounds them (asserted).  Per-process count, last abort code and wall time go into the test record
    (record_property) and DESIGN.md.  (Round 6: also at n = 10^4, where every process runs the PERSISTENT launch with its tasks
    dealt to the XCDs -- three launches of 2040 workgroups compete for the chip's slots, so an XCD may hold few workgroups of
    one of them: the classes of that launch must still be carried by the others.)"""
    import torch.multiprocessing as mp
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    world = 3
    mp.spawn(_optim_worker, args=(world, _free_port(), g, nevals, str(tmp_path)), nprocs=world, join=True)
    locs, X, th, z = _grid_problem(g)
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
    import time
    t0 = time.perf_counter()
    for rank in range(world):
        res = np.load(os.path.join(str(tmp_path), "worker%d.npy" % rank))
        for e in range(nevals):
            t = {k: np.array(v, dtype=float).copy() for k, v in th.items()}
            t["std.dev"][1] += 0.01 * (rank * nevals + e)
            want = fit.neg2loglik_core(t)[0]
            assert abs(res[e] - want) <= 1e-10 * abs(want), (rank, e)
        line = ("%d evaluations in %.1f ms, engine time-outs %d (last code 0x%x), engine active at the end: %d"
                % (nevals, 1e3 * res[nevals + 3], int(res[nevals]), int(res[nevals + 1]), int(res[nevals + 2])))
        print("worker %d: %s" % (rank, line))
        record_property("worker%d" % rank, line)              # (pytest -q swallows the print; the junit / driver record keeps this)
        # the back-off bounds what a process can lose: a time-out sends the next 2, 4, ... operations to the plain schedule,
        # so eleven operations hold at most three
        assert int(res[nevals]) <= 3, (rank, line)
    single = time.perf_counter() - t0
    print("the same %d evaluations from ONE process, one after the other: %.1f ms" % (world * nevals, 1e3 * single))
    assert fit.engine_state()["retries"] == 0            # alone on the device the engine never times out
    fit.close()
