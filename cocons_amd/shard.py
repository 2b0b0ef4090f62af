"""Row-block sharded -2 log-likelihood across the GPUs of one node.

The reference's `chol` (R/neg2loglikelihood.R:200) walks the upper triangle of Sigma row block by row block.
Here Sigma is ROW-BLOCK partitioned: block b = rows 256 b .. 256 b + 255, dealt over the ranks in
groups of G consecutive blocks, owner(b) = (b div G) mod world (COCONS_SHARD_GROUP, default 4).  A rank assembles,
solves and updates ITS rows of every column:

    for each 256-column block k:
        owner(k): factor the 256 x 256 diagonal block (every earlier update of its rows is local)
        broadcast L_kk from owner(k)                                           <- 0.5 MB, on the chain
        every rank: solve ITS rows of the panel, X = B L_kk^-T
        owner(k+1): update its diagonal block (k+1,k+1) with its own rows of X, factor it, start its broadcast
                    (issued in front of the all-gather below, on a stream and communicator of its own: the chain
                    diagonal block -> diagonal block never waits for the bulk exchange)
        all-gather of the solved rows, packed by owner                         <- B_k / world per rank
        every rank: update ITS rows of the trailing matrix with the gathered panel   (MFMA fp64)
    all-reduce of {sum log diag, Gram of the rhs rows} (1 + r^2 doubles) and of the failing minor

The schedule, the RCCL collectives (on a communication stream of their own) and the final all-reduce all live
INSIDE the HIP library (`sharded_eval` in csrc/api.hip): once a fit has collectives, `cocons_neg2loglik_dense`
on it is the sharded evaluation.  This module only wires a fit to its communicator:

  * `ShardedFit.init_rccl(dist)`    one process per GPU: rank 0 draws the RCCL unique id
    (`cocons_comm_unique_id`), the 128 bytes travel over the host's process group
    (torch.distributed here; R would use its own socket/MPI), every rank calls
    `cocons_fit_comm_init`.  torch carries no panel, no reduction, no stream.
  * `ShardedFit.init_host_transport(dist)`   tests: several ranks share ONE GPU, which RCCL refuses;
    the library's broadcast / all-gather / all-reduce hooks (`cocons_fit_set_collectives`,
    `cocons_fit_set_allgather`) are served by gloo through host memory.  Same native schedule, different wire.
  * `MultiFit`   one process, several GPUs (`cocons_multi_*`, ncclCommInitAll).

`sharded_neg2loglik_core(engine, ...)` further down is the same schedule in Python over an abstract
engine: the CPU tests run it over gloo with a numpy engine (tests/np_shard_engine.py), which pins the
schedule's logic (ownership, look-ahead order, owner-packed exchange) where no GPU exists.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

from . import _lib
from .host import CoconsFit, _p, theta_table

INFO_OK = 0x7F7F7F7F
LOG_2PI = math.log(2.0 * math.pi)


def _hip_runtime():
    """libamdhip64 through ctypes, for the host-transport hooks (device <-> host copies)."""
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6", "/opt/rocm/lib/libamdhip64.so"):
        try:
            lib = ctypes.CDLL(name)
            break
        except OSError:
            lib = None
    if lib is None:
        raise _lib.CoconsHipError("cannot load the HIP runtime")
    lib.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    lib.hipMemcpy.restype = ctypes.c_int
    lib.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    lib.hipStreamSynchronize.restype = ctypes.c_int
    return lib


class ShardedFit(CoconsFit):
    """CoconsFit whose evaluation is split over the ranks of a process group; see the module
    docstring.  `neg2loglik_core` (inherited) returns the same value on every rank."""

    def __init__(self, locs, x_covariates, z, smooth_limits, device):
        super().__init__(locs, x_covariates, z, smooth_limits, device=device)
        self._keep = []          # ctypes callbacks must outlive the handle

    def world(self):
        return int(self._L.cocons_fit_world(self._h))

    def init_rccl(self, dist, rank, world, group=None):
        """One process per GPU: share the RCCL unique id over `dist` (any backend that moves 128 host
        bytes) and create this rank's communicator inside the library."""
        import torch
        buf = (ctypes.c_ubyte * _lib.UNIQUE_ID_BYTES)()
        if rank == 0:
            _lib.check(self._L.cocons_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)), "cocons_comm_unique_id")
        t = torch.tensor(list(buf), dtype=torch.uint8)
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.broadcast(t, src=0, group=group)
        raw = bytes(t.cpu().tolist())
        idb = (ctypes.c_ubyte * _lib.UNIQUE_ID_BYTES).from_buffer_copy(raw)
        _lib.check(self._L.cocons_fit_comm_init(self._h, world, rank, ctypes.cast(idb, ctypes.c_void_p)),
                   "cocons_fit_comm_init")

    def comm_info(self):
        """{"count": ncclCommCount, "rank": ncclCommUserRank, "device": ncclCommCuDevice} of this fit's communicator
        (world / rank / device for a caller-provided transport; count 0 without collectives)."""
        c, u, d = ctypes.c_int(0), ctypes.c_int(-1), ctypes.c_int(-1)
        _lib.check(self._L.cocons_fit_comm_info(self._h, ctypes.byref(c), ctypes.byref(u), ctypes.byref(d)),
                   "cocons_fit_comm_info")
        return {"count": c.value, "rank": u.value, "device": d.value}

    def init_host_transport(self, dist, rank, world, group=None):
        """Tests: serve the library's broadcast / all-reduce hooks with `dist` (gloo) through host memory."""
        import torch
        hip = _hip_runtime()
        D2H, H2D = 2, 1

        def bcast(user, dev_ptr, nbytes, root, stream):
            try:
                if hip.hipStreamSynchronize(stream) != 0:
                    return 1
                host = np.empty(nbytes // 8, dtype=np.float64)
                if rank == root and hip.hipMemcpy(host.ctypes.data, dev_ptr, nbytes, D2H) != 0:
                    return 2
                dist.broadcast(torch.from_numpy(host), src=root, group=group)
                if rank != root and hip.hipMemcpy(dev_ptr, host.ctypes.data, nbytes, H2D) != 0:
                    return 3
                return 0
            except Exception:                                   # noqa: BLE001
                return 9

        def allreduce(user, ptr, count, op):
            try:
                arr = np.ctypeslib.as_array(ptr, shape=(count,))
                t = torch.from_numpy(arr.copy())
                dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MIN, group=group)
                arr[:] = t.numpy()
                return 0
            except Exception:                                   # noqa: BLE001
                return 9

        def allgather(user, dev_ptr, nbytes, stream):
            # dev_ptr: `world` slots of nbytes; slot `rank` is this rank's contribution, every other slot is filled here
            try:
                if hip.hipStreamSynchronize(stream) != 0:
                    return 1
                mine = np.empty(nbytes // 8, dtype=np.float64)
                if hip.hipMemcpy(mine.ctypes.data, dev_ptr + rank * nbytes, nbytes, D2H) != 0:
                    return 2
                parts = [torch.empty(nbytes // 8, dtype=torch.float64) for _ in range(world)]
                dist.all_gather(parts, torch.from_numpy(mine), group=group)
                for r in range(world):
                    if r != rank and hip.hipMemcpy(dev_ptr + r * nbytes, parts[r].numpy().ctypes.data, nbytes, H2D) != 0:
                        return 3
                return 0
            except Exception:                                   # noqa: BLE001
                return 9

        cb_b, cb_a, cb_g = _lib.BCAST_FN(bcast), _lib.ALLREDUCE_FN(allreduce), _lib.ALLGATHER_FN(allgather)
        self._keep += [cb_b, cb_a, cb_g]
        _lib.check(self._L.cocons_fit_set_collectives(self._h, rank, world, ctypes.cast(cb_b, ctypes.c_void_p),
                                                      ctypes.cast(cb_a, ctypes.c_void_p), None),
                   "cocons_fit_set_collectives")
        _lib.check(self._L.cocons_fit_set_allgather(self._h, ctypes.cast(cb_g, ctypes.c_void_p)), "cocons_fit_set_allgather")


class MultiFit:
    """One process driving several GPUs (cocons_multi_*): what a single R session calls."""

    def __init__(self, locs, x_covariates, z, smooth_limits, devices):
        L = _lib.load()
        self._L = L
        locs = np.asfortranarray(np.asarray(locs, dtype=np.float64))
        X = np.asfortranarray(np.asarray(x_covariates, dtype=np.float64))
        self.n, self.p = X.shape
        z = np.asfortranarray(np.asarray(z, dtype=np.float64).reshape(self.n, -1))
        self.r = z.shape[1]
        sl = np.asarray(smooth_limits, dtype=np.float64)
        dev = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        self._h = L.cocons_multi_create(self.n, self.p, self.r, _p(locs), _p(X), _p(z), _p(sl), len(devices), dev)
        if not self._h:
            raise _lib.CoconsHipError("cocons_multi_create failed: " + _lib.last_error())

    def neg2loglik_core(self, theta_list):
        T = theta_table(theta_list)
        mean = np.ascontiguousarray(np.asarray(theta_list["mean"], dtype=np.float64))
        val = ctypes.c_double(0.0)
        parts = np.zeros(1 + self.r)
        _lib.check(self._L.cocons_multi_neg2loglik_dense(self._h, _p(T), _p(mean), ctypes.byref(val), _p(parts)),
                   "cocons_multi_neg2loglik_dense")
        return val.value, parts

    def neg2loglik_batch_core(self, theta_lists):
        """Independent evaluations dealt over the handle's devices (replica mode, cocons_multi_neg2loglik_batch):
        (values, status) like CoconsFit.neg2loglik_batch_core."""
        nb = len(theta_lists)
        T = np.ascontiguousarray(np.stack([theta_table(t) for t in theta_lists], axis=0)) if nb else np.zeros((0, 6, self.p))
        M = np.ascontiguousarray(np.stack([np.asarray(t["mean"], dtype=np.float64) for t in theta_lists], axis=0)) \
            if nb else np.zeros((0, self.p))
        vals = np.zeros(nb)
        st = np.zeros(nb, dtype=np.int32)
        _lib.check(self._L.cocons_multi_neg2loglik_batch(self._h, nb, _p(T), _p(M), _p(vals),
                                                         st.ctypes.data_as(ctypes.POINTER(ctypes.c_int))),
                   "cocons_multi_neg2loglik_batch")
        return vals, st

    def comm_ranks(self):
        """(devices of the handle, ncclCommCount of its communicators; 0 = none)."""
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self._L.cocons_multi_comm_ranks(self._h, ctypes.byref(a), ctypes.byref(b)), "cocons_multi_comm_ranks")
        return a.value, b.value

    def predict_core(self, theta_list, locs_pred, x_covariates_pred, z_col=0):
        """(stochastic, quadform) of cocoPredict's dense core, the prediction locations split over the devices."""
        T = theta_table(theta_list)
        mean = np.ascontiguousarray(np.asarray(theta_list["mean"], dtype=np.float64))
        lp = np.asfortranarray(np.asarray(locs_pred, dtype=np.float64))
        Xp = np.asfortranarray(np.asarray(x_covariates_pred, dtype=np.float64))
        m = Xp.shape[0]
        st, qf = np.empty(m), np.empty(m)
        _lib.check(self._L.cocons_multi_predict_dense(self._h, _p(T), _p(mean), int(z_col), m, _p(lp), _p(Xp),
                                                      _p(st), _p(qf)), "cocons_multi_predict_dense")
        return st, qf

    def close(self):
        if getattr(self, "_h", None):
            self._L.cocons_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:                                       # noqa: BLE001
            pass


def sharded_neg2loglik_core(engine, theta_list, dist, rank, world, group=None, lookahead=True):
    """The row-block sharded schedule over an abstract engine: the Python twin of `sharded_eval` in csrc/api.hip, run by
    the CPU tests with a numpy engine.  Engine interface: begin / num_blocks / owner / factor_diag / diag_tensor /
    set_diag / solve / ahead / exchanges / pack / set_gathered / update / finish / make_tensor.
    Returns (sum_logliks, parts), identical on every rank; raises CholeskyError on every rank if any pivot failed.

    Look-ahead (the library's order): the owner of block k+1 updates its diagonal block with its OWN solved rows and
    factors it before the bulk exchange of step k, and the broadcast of L_(k+1,k+1) is issued IN FRONT of the all-gather
    of step k (round 5; the library gives it a stream and a communicator of its own) -- the chain of diagonal blocks waits
    neither for a trailing update nor for a bulk exchange.  lookahead=False: every step in plain
    order (factor | broadcast | solve | all-gather | update), the same arithmetic."""
    engine.begin(theta_list, rank, world)
    nb = engine.num_blocks()
    owner_of = engine.owner

    def bcast_diag(k):
        if world > 1:
            dist.broadcast(engine.diag_tensor(k), src=owner_of(k), group=group)
        if rank != owner_of(k):
            engine.set_diag(k)

    def gather(k):
        mine = engine.pack(k)
        if world > 1:
            parts = [mine.new_empty(mine.shape) for _ in range(world)]
            dist.all_gather(parts, mine, group=group)
        else:
            parts = [mine]
        engine.set_gathered(k, parts)

    if rank == owner_of(0):
        engine.factor_diag(0)
    if lookahead:
        bcast_diag(0)
    for k in range(nb):
        if not lookahead:
            if k > 0 and rank == owner_of(k):
                engine.factor_diag(k)
            bcast_diag(k)
        engine.solve(k)                                   # this rank's rows below block k
        ahead = lookahead and k + 1 < nb and engine.exchanges(k)
        if ahead and rank == owner_of(k + 1):
            engine.ahead(k + 1)                           # own rows of X: local
            engine.factor_diag(k + 1)
        if lookahead and k + 1 < nb:
            bcast_diag(k + 1)
        if engine.exchanges(k):
            gather(k)
        if engine.exchanges(k):
            engine.update(k, skip_diag=(k + 1 if (ahead and rank == owner_of(k + 1)) else None))
    part, info = engine.finish()
    if world > 1:
        t = engine.make_tensor(np.concatenate([part, [-float(info)]]))
        red = t[:-1]
        dist.all_reduce(red, op=dist.ReduceOp.SUM, group=group)
        mx = t[-1:]
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)     # max of -info = -min(info)
        part = red.cpu().numpy()
        info = int(-mx.cpu().item())
    if info != INFO_OK:
        raise _lib.CholeskyError(min(info, engine.n))
    r = engine.r
    logdet = part[0]
    total = 0.0
    parts = np.zeros(1 + r)
    parts[0] = logdet
    for c in range(r):
        quad = part[1 + c * r + c]
        total += engine.n * LOG_2PI + 2 * logdet + quad
        parts[1 + c] = quad
    return total, parts


def sharded_predict_core(fit, theta_list, locs_pred, x_covariates_pred, dist, rank, world, group=None, z_col=0):
    """Dense kriging (R/predict.R:136-183) with the m prediction locations split over the ranks
    (BASELINE config C5: the right-hand sides shard): every rank factors Sigma with
    its own m/world cross-covariance rows as border -- no exchange during the solve -- and the
    (stochastic, quadratic-form) vectors are gathered at the end (O(m) doubles).  `fit` is a plain
    CoconsFit per rank."""
    lp = np.asarray(locs_pred, dtype=np.float64)
    Xp = np.asarray(x_covariates_pred, dtype=np.float64)
    m = lp.shape[0]
    lo, hi = (m * rank) // world, (m * (rank + 1)) // world
    st = np.zeros(m)
    qf = np.zeros(m)
    if hi > lo:
        s, q = fit.predict_core(theta_list, lp[lo:hi], Xp[lo:hi], z_col=z_col)
        st[lo:hi], qf[lo:hi] = s, q
    if world > 1:
        import torch
        t = torch.from_numpy(np.stack([st, qf]))
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)     # disjoint supports: sum = concatenation
        st, qf = t[0].cpu().numpy(), t[1].cpu().numpy()
    return st, qf
