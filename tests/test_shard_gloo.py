"""N > 1 path on CPU: the production schedule (cocons_amd.shard.sharded_neg2loglik_core)
driven over gloo with world_size 2 and 3, using the numpy test engine.  Checks the result
against the dense single-process oracle value and that every rank returns the same number."""
import math
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, seed, out_dir, lookahead=True, group=1):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from cocons_amd import workloads as wl
    from cocons_amd.shard import sharded_neg2loglik_core
    from np_shard_engine import NumpyShardEngine
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(seed)
    locs = rng.uniform(0, 1, size=(n, 2))
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full(scale0=np.log(0.2))
    th["mean"] = np.array([0.1, -0.2, 0.05])
    z = rng.standard_normal((n, 2))
    eng = NumpyShardEngine(O, locs, X, z, wl.SMOOTH_LIMITS, group=group)
    val, parts = sharded_neg2loglik_core(eng, th, dist, rank, world, lookahead=lookahead)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.concatenate([[val], parts]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,lookahead,group", [(2, 700, True, 1), (3, 900, True, 1), (2, 200, True, 1),
                                                        (2, 700, False, 1), (4, 1300, True, 1), (4, 200, True, 1),
                                                        (3, 100, False, 1),
                                                        # blocks dealt in groups of consecutive blocks (the library's
                                                        # default deal): 6 blocks over 4 ranks in pairs leaves a rank idle
                                                        (2, 1300, True, 2), (4, 1300, True, 2), (3, 900, False, 3)])
def test_sharded_schedule_over_gloo(oracle, tmp_path, world, n, lookahead, group):
    import torch.multiprocessing as mp
    from cocons_amd import workloads as wl
    seed = 100 + n
    mp.spawn(_worker, args=(world, _free_port(), n, seed, str(tmp_path), lookahead, group), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), "rank%d.npy" % r)) for r in range(world)]
    for r in res[1:]:
        assert np.array_equal(r, res[0])              # identical on every rank
    rng = np.random.default_rng(seed)
    locs = rng.uniform(0, 1, size=(n, 2))
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full(scale0=np.log(0.2))
    th["mean"] = np.array([0.1, -0.2, 0.05])
    z = rng.standard_normal((n, 2))
    pp = wl.par_pos_full()
    pp["mean"] = [True] * 3
    tv = np.concatenate([th["mean"], wl.theta_vector_from_lists(th, wl.par_pos_full())])
    want = oracle.GetNeg2loglikelihood(tv, pp, locs, X, wl.SMOOTH_LIMITS, z, n, (0, 0, 0))
    assert np.isfinite(res[0][0])
    assert abs(res[0][0] - want) < 1e-10 * abs(want)


def test_sharded_schedule_world1_matches(oracle):
    """world = 1 degenerates to the plain blocked factorisation (no collective)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cocons_amd import workloads as wl
    from cocons_amd.shard import sharded_neg2loglik_core
    from np_shard_engine import NumpyShardEngine
    rng = np.random.default_rng(3)
    n = 300
    locs = rng.uniform(0, 1, size=(n, 2))
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full(scale0=np.log(0.2))
    z = rng.standard_normal(n)
    eng = NumpyShardEngine(oracle, locs, X, z, wl.SMOOTH_LIMITS)
    val, parts = sharded_neg2loglik_core(eng, th, None, 0, 1)
    S = oracle.cov_rns(th, locs, X, wl.SMOOTH_LIMITS)
    info, ld, quad, _ = oracle.chol_ld(S, z)
    truth = n * math.log(2 * math.pi) + 2 * ld + quad[0]
    assert abs(val - truth) < 1e-11 * abs(truth)


def _predict_worker(rank, world, port, out_dir):
    """sharded_predict_core over gloo with a fake fit whose predict_core is the oracle (CPU)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from cocons_amd import workloads as wl
    from cocons_amd.shard import sharded_predict_core
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(4)
    n, m = 120, 37
    locs = rng.uniform(0, 1, size=(n, 2))
    X = wl.design_from_locs(locs)["std.covs"]
    lp = rng.uniform(0, 1, size=(m, 2))
    Xp = wl.design_from_locs(lp)["std.covs"]
    th = wl.theta_full(scale0=np.log(0.2))
    z = rng.standard_normal(n)

    class OracleFit:
        def predict_core(self, tl, lpp, Xpp, z_col=0):
            S = O.cov_rns(tl, locs, X, wl.SMOOTH_LIMITS)
            C = O.cov_rns_pred(tl, locs, lpp, X, Xpp, wl.SMOOTH_LIMITS)
            sol = np.linalg.solve(S, C.T)
            return (z - X @ tl["mean"]) @ sol, np.sum(C * sol.T, axis=1)

    st, qf = sharded_predict_core(OracleFit(), th, lp, Xp, dist, rank, world)
    full = OracleFit().predict_core(th, lp, Xp)
    assert np.allclose(st, full[0], rtol=1e-12) and np.allclose(qf, full[1], rtol=1e-12)
    np.save(os.path.join(out_dir, "p%d.npy" % rank), st)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_predict_over_gloo(oracle, tmp_path):
    """C5's multi-GPU shape: prediction locations split over ranks, results all-gathered."""
    import torch.multiprocessing as mp
    mp.spawn(_predict_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    a = [np.load(os.path.join(str(tmp_path), "p%d.npy" % r)) for r in range(3)]
    assert np.array_equal(a[0], a[1]) and np.array_equal(a[0], a[2])
