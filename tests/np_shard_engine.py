"""TEST DOUBLE (not product code): a numpy implementation of the engine interface that
cocons_amd.shard.sharded_neg2loglik_core drives, with the same storage conventions as the HIP path (column-major lower
factor, 128-tiles, 256-row blocks dealt over the ranks in groups of `group`, rhs rows under the matrix, the solved rows
exchanged in an owner-packed buffer with 64-row tiles).  Lets the N > 1 schedule and its collectives run over gloo on CPU.
Rows a rank does not own are filled with NaN, so any use of data that was never exchanged shows up in the result."""
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
        self.mt = self.nt + 1                       # one tile row of right-hand sides under the matrix
        self.lda = self.mt * TILE
        self.info = 0x7F7F7F7F

    def num_blocks(self):
        return (self.nt + PT - 1) // PT

    def owner(self, b):
        """(b div G) mod world: csrc/api.hip shard_owner."""
        return (b // self.group) % self.world

    def _owner64(self, ti):
        return self.owner(ti // 4)

    def _cols(self, k):
        c0 = k * PT * TILE
        return c0, min(c0 + PT * TILE, self.npad)

    def exchanges(self, k):
        """Is there a trailing matrix right of block k (otherwise only right-hand-side rows lie below: no exchange)."""
        return self._cols(k)[1] < self.npad

    def _own_tiles(self, k, rank=None):
        rank = self.rank if rank is None else rank
        c1 = self._cols(k)[1]
        return [ti for ti in range(c1 // 64, 2 * self.mt) if self._owner64(ti) == rank]

    def begin(self, theta_list, rank, world):
        self.rank, self.world = rank, world
        S = np.tril(self.O.cov_rns(theta_list, self.locs, self.X, self.sl))
        A = np.full((self.lda, self.npad), np.nan)
        resid = self.z - (self.X @ np.asarray(theta_list["mean"], float))[:, None]
        full = np.zeros((self.lda, self.npad))
        full[: self.n, : self.n] = S
        for c in range(self.n, self.npad):
            full[c, c] = 1.0
        full[self.npad: self.npad + self.r, : self.n] = resid.T
        for ti in range(2 * self.mt):
            if self._owner64(ti) == rank:
                A[64 * ti: 64 * ti + 64] = full[64 * ti: 64 * ti + 64]
        self.A = A
        self.L = {}

    def factor_diag(self, k):
        c0, c1 = self._cols(k)
        blk = np.tril(self.A[c0:c1, c0:c1])
        blk = blk + np.tril(blk, -1).T
        try:
            L = np.linalg.cholesky(blk)
        except np.linalg.LinAlgError:
            self.info = min(self.info, c0 + 1)
            L = np.eye(c1 - c0)
        self.A[c0:c1, c0:c1] = L
        self.L[k] = torch.from_numpy(np.ascontiguousarray(L))

    def diag_tensor(self, k):
        c0, c1 = self._cols(k)
        if k not in self.L:
            self.L[k] = torch.full((c1 - c0, c1 - c0), float("nan"), dtype=torch.float64)
        return self.L[k]

    def set_diag(self, k):
        c0, c1 = self._cols(k)
        self.A[c0:c1, c0:c1] = self.L[k].numpy()

    def solve(self, k):
        from scipy.linalg import solve_triangular
        c0, c1 = self._cols(k)
        L = self.L[k].numpy()
        for ti in self._own_tiles(k):
            r0 = 64 * ti
            self.A[r0:r0 + 64, c0:c1] = solve_triangular(L, self.A[r0:r0 + 64, c0:c1].T, lower=True).T

    def ahead(self, j):
        """Owner of block j: its diagonal block updated with its OWN solved rows of panel j - 1."""
        c0, c1 = self._cols(j - 1)
        d0, d1 = self._cols(j)
        Xo = self.A[d0:d1, c0:c1]
        self.A[d0:d1, d0:d1] -= Xo @ Xo.T

    def _slot_rows(self, k):
        return 64 * max(len(self._own_tiles(k, r)) for r in range(self.world))

    def pack(self, k):
        c0, c1 = self._cols(k)
        out = np.zeros((self._slot_rows(k), c1 - c0))
        for pos, ti in enumerate(self._own_tiles(k)):
            out[64 * pos: 64 * pos + 64] = self.A[64 * ti: 64 * ti + 64, c0:c1]
        return torch.from_numpy(out)

    def set_gathered(self, k, parts):
        c0, c1 = self._cols(k)
        P = np.full((self.lda, c1 - c0), np.nan)
        for r in range(self.world):
            slot = parts[r].numpy()
            for pos, ti in enumerate(self._own_tiles(k, r)):
                P[64 * ti: 64 * ti + 64] = slot[64 * pos: 64 * pos + 64]
        self.P = P

    def update(self, k, skip_diag=None):
        c0, c1 = self._cols(k)
        for ti in self._own_tiles(k):
            r0 = 64 * ti
            ce = min(r0 + 64, self.npad)                  # columns up to the tile's own diagonal
            if ce <= c1:
                continue
            cols = np.arange(c1, ce)
            if skip_diag is not None:
                d0, d1 = self._cols(skip_diag)
                if d0 <= r0 < d1:                         # a row of that diagonal block proper (not a right-hand-side row)
                    cols = cols[(cols < d0) | (cols >= d1)]
            if cols.size:
                self.A[r0:r0 + 64, cols] -= self.P[r0:r0 + 64] @ self.P[cols].T

    def finish(self):
        part = np.zeros(1 + self.r * self.r)
        if self._owner64(2 * self.nt) == self.rank:       # the rank that owns the right-hand-side rows has every L_kk too
            part[0] = np.sum(np.log(np.diag(self.A)[: self.n]))
            Y = self.A[self.npad: self.npad + self.r, : self.n]
            part[1:] = (Y @ Y.T).ravel()
        return part, self.info

    def make_tensor(self, arr):
        return torch.as_tensor(arr)
