"""Column-panel sharded -2 log-likelihood across the GPUs of one node.

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI) as plumbing.
The reference's `chol` (R/neg2loglikelihood.R:200) reads the upper triangle of Sigma row
block by row block; those row blocks are the column panels of the lower factor kept on
the device, so "Sigma row-block partitioned" = panels dealt block-cyclically over ranks:

    for each panel k (256 columns):
        owner(k) = k mod world : factor the panel in place (potrf + panel solve), pack it
        broadcast the packed panel (<= 20 MB at n = 10^4) from its owner      <- the only collective
        every rank: update its OWN panels right of k with the received panel   (MFMA fp64)
    all-reduce of {sum log diag, Gram of the rhs rows} partial sums           (1 + r^2 doubles)

Assembly needs no exchange: every rank builds the per-location vectors (O(n p)) and
assembles only its own panels.  The right-hand sides ride along as extra rows of every
panel, so no distributed triangular solve exists.

The schedule below is engine-agnostic: `engine` is a `ShardedFit` (HIP) in production; the
CPU tests drive the same loop over gloo with a numpy engine that lives under tests/.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

from . import _lib
from .host import CoconsFit, _p, theta_table

INFO_OK = 0x7F7F7F7F
LOG_2PI = math.log(2.0 * math.pi)


class ShardedFit(CoconsFit):
    """CoconsFit whose factorisation is split over the ranks of a process group.  The
    exchange buffers are torch tensors so that torch.distributed can broadcast them in
    place, and all kernels run on torch's current stream (collectives are ordered
    against it by torch)."""

    def __init__(self, locs, x_covariates, z, smooth_limits, device):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        super().__init__(locs, x_covariates, z, smooth_limits, device=device)
        L = self._L
        nbytes = int(L.cocons_shard_exchange_bytes(self._h))
        self._xbuf = [torch.empty(nbytes // 8, dtype=torch.float64, device=self.device) for _ in range(2)]
        self._pinned_stream = torch.cuda.current_stream().cuda_stream
        _lib.check(L.cocons_fit_set_stream(self._h, ctypes.c_void_p(self._pinned_stream)),
                   "cocons_fit_set_stream")
        _lib.check(L.cocons_shard_set_exchange(self._h, ctypes.c_void_p(self._xbuf[0].data_ptr()),
                                               ctypes.c_void_p(self._xbuf[1].data_ptr()), nbytes),
                   "cocons_shard_set_exchange")

    # engine interface ------------------------------------------------------
    def begin(self, theta_list, rank, world):
        # (re)pin the handle to the stream that is current NOW: torch orders the collectives of this
        # evaluation against its current stream, so the kernels must be on the same one
        cur = self.torch.cuda.current_stream().cuda_stream
        if cur != self._pinned_stream:
            _lib.check(self._L.cocons_fit_set_stream(self._h, ctypes.c_void_p(cur)), "cocons_fit_set_stream")
            self._pinned_stream = cur
        T = theta_table(theta_list)
        mean = np.ascontiguousarray(np.asarray(theta_list["mean"], dtype=np.float64))
        _lib.check(self._L.cocons_shard_begin(self._h, _p(T), _p(mean), rank, world), "cocons_shard_begin")

    def num_panels(self):
        return int(self._L.cocons_shard_num_panels(self._h))

    def panel_factor(self, k):
        _lib.check(self._L.cocons_shard_panel_factor(self._h, k), "cocons_shard_panel_factor")

    def panel_tensor(self, k):
        nbytes = ctypes.c_longlong(0)
        ptr = ctypes.c_void_p()
        _lib.check(self._L.cocons_shard_panel_buffer(self._h, k, ctypes.byref(ptr), ctypes.byref(nbytes)),
                   "cocons_shard_panel_buffer")
        buf = self._xbuf[k & 1]
        assert ptr.value == buf.data_ptr()
        return buf[: nbytes.value // 8]

    def panel_apply(self, k, j0=None, j1=None):
        """update own panels j in [j0, j1) (default: all right of k) with the received panel k"""
        j0 = k + 1 if j0 is None else j0
        j1 = -1 if j1 is None else j1
        _lib.check(self._L.cocons_shard_panel_apply_range(self._h, k, j0, j1), "cocons_shard_panel_apply_range")

    def finish(self):
        part = np.zeros(1 + self.r * self.r)
        info = ctypes.c_int(0)
        _lib.check(self._L.cocons_shard_finish(self._h, _p(part), ctypes.byref(info)), "cocons_shard_finish")
        return part, info.value

    def make_tensor(self, arr):
        return self.torch.as_tensor(arr, device=self.device)


def sharded_neg2loglik_core(engine, theta_list, dist, rank, world, group=None, lookahead=True):
    """One sharded evaluation.  Returns (sum_logliks, parts) like CoconsFit.neg2loglik_core,
    identical on every rank; raises CholeskyError on every rank if any panel failed.

    Look-ahead: the owner of panel k+1 updates and factors that panel FIRST and starts its
    broadcast asynchronously; every rank then applies panel k to the rest of its panels while
    the broadcast of k+1 is in flight (the exchange buffers alternate, so the receive of k+1
    never touches the buffer panel k is being read from)."""
    engine.begin(theta_list, rank, world)
    npan = engine.num_panels()
    if world == 1:
        for k in range(npan):
            engine.panel_factor(k)
            engine.panel_apply(k)
    elif not lookahead:
        for k in range(npan):
            owner = k % world
            if rank == owner:
                engine.panel_factor(k)
            dist.broadcast(engine.panel_tensor(k), src=owner, group=group)
            engine.panel_apply(k)
    else:
        if rank == 0:
            engine.panel_factor(0)
        work = dist.broadcast(engine.panel_tensor(0), src=0, group=group, async_op=True)
        for k in range(npan):
            work.wait()                                   # panel k is in its exchange buffer
            nxt = k + 1
            if nxt < npan:
                if rank == nxt % world:
                    engine.panel_apply(k, nxt, nxt + 1)   # only the columns of panel k+1 ...
                    engine.panel_factor(nxt)              # ... factor it ...
                # ... and put it on the wire while everybody applies panel k to the rest
                work = dist.broadcast(engine.panel_tensor(nxt), src=nxt % world, group=group, async_op=True)
                if rank == nxt % world:
                    engine.panel_apply(k, nxt + 1, None)
                else:
                    engine.panel_apply(k)
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
    (stochastic, quadratic-form) vectors are all-gathered.  `fit` is a plain CoconsFit per rank."""
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
