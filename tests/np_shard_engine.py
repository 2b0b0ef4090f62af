"""TEST DOUBLE (not product code): a numpy implementation of the engine interface that
cocons_amd.shard.sharded_neg2loglik_core drives, with the same storage conventions as the
HIP engine (column-major lower factor, 128-tiles, 256-column panels dealt over the ranks in groups of `group`,
rhs rows under the matrix, packed exchange buffers).  Lets the N>1 schedule and its
collectives run over gloo on CPU.  Panels a rank does not own are filled with NaN, so any
use of data that was never broadcast shows up in the result."""
import numpy as np
import torch

TILE, PT = 128, 2


class NumpyShardEngine:
    def __init__(self, oracle, locs, X, z, smooth_limits, group=1):
        self.O = oracle
        self.group = max(1, int(group))
        self.locs, self.X = np.asarray(locs, float), np.asarray(X, float)
        self.z = np.asarray(z, float).reshape(self.X.shape[0], -1)
        self.n, self.r = self.z.shape
        self.sl = smooth_limits
        self.npad = (self.n + TILE - 1) // TILE * TILE
        self.nt = self.npad // TILE
        self.lda = self.npad + TILE
        self.xbuf = [torch.zeros(self.lda * PT * TILE, dtype=torch.float64) for _ in range(2)]
        self.info = 0x7F7F7F7F

    def num_panels(self):
        return (self.nt + PT - 1) // PT

    def owner(self, k):
        """(k div G) mod world: csrc/api.hip shard_owner."""
        return (k // self.group) % self.world

    def _cols(self, k):
        c0 = k * PT * TILE
        return c0, min(c0 + PT * TILE, self.npad)

    def begin(self, theta_list, rank, world):
        self.rank, self.world = rank, world
        S = self.O.cov_rns(theta_list, self.locs, self.X, self.sl)
        A = np.full((self.lda, self.npad), np.nan)
        resid = self.z - (self.X @ np.asarray(theta_list["mean"], float))[:, None]
        for k in range(self.num_panels()):
            if self.owner(k) != rank:
                continue
            c0, c1 = self._cols(k)
            A[:, c0:c1] = 0.0
            hi = min(c1, self.n)
            if hi > c0:
                A[: self.n, c0:hi] = np.tril(S)[:, c0:hi]
                A[self.npad: self.npad + self.r, c0:hi] = resid[c0:hi].T
            for c in range(max(c0, self.n), c1):
                A[c, c] = 1.0
        self.A = A

    def panel_factor(self, k):
        c0, c1 = self._cols(k)
        A = self.A
        blk = np.tril(A[c0:c1, c0:c1])
        blk = blk + np.tril(blk, -1).T
        try:
            L = np.linalg.cholesky(blk)
        except np.linalg.LinAlgError:
            self.info = min(self.info, c0 + 1)
            L = np.eye(c1 - c0)
        A[c0:c1, c0:c1] = L
        from scipy.linalg import solve_triangular
        A[c1:, c0:c1] = solve_triangular(L, A[c1:, c0:c1].T, lower=True).T
        rows = self.lda - c0
        self.xbuf[k & 1][: rows * (c1 - c0)] = torch.from_numpy(
            np.asfortranarray(A[c0:, c0:c1]).ravel(order="F").copy())

    def panel_tensor(self, k):
        c0, c1 = self._cols(k)
        return self.xbuf[k & 1][: (self.lda - c0) * (c1 - c0)]

    def panel_apply(self, k, j0=None, j1=None):
        c0, c1 = self._cols(k)
        rows = self.lda - c0
        P = self.panel_tensor(k).numpy().reshape((rows, c1 - c0), order="F")
        j0 = k + 1 if j0 is None else max(j0, k + 1)
        j1 = self.num_panels() if (j1 is None or j1 < 0) else min(j1, self.num_panels())
        for j in range(j0, j1):
            if self.owner(j) != self.rank:
                continue
            d0, d1 = self._cols(j)
            # rows >= d0 of own panel j:  C -= P(rows) P(cols d0:d1)^T
            self.A[d0:, d0:d1] -= P[d0 - c0:, :] @ P[d0 - c0: d1 - c0, :].T

    def finish(self):
        part = np.zeros(1 + self.r * self.r)
        for k in range(self.num_panels()):
            if self.owner(k) != self.rank:
                continue
            c0, c1 = self._cols(k)
            hi = min(c1, self.n)
            if hi <= c0:
                continue
            part[0] += np.sum(np.log(np.diag(self.A)[c0:hi]))
            Y = self.A[self.npad: self.npad + self.r, c0:hi]
            part[1:] += (Y @ Y.T).ravel()
        return part, self.info

    def make_tensor(self, arr):
        return torch.as_tensor(arr)
