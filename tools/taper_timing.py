"""Time the taper objective (GetNeg2loglikelihoodTaper through the band-limited dense-tile factorisation on the device)
on a g x g grid with a Wendland-1 taper of range delta: python tools/taper_timing.py [g=100] [delta=0.06] [cpu].
The CPU comparison (SuperLU) runs for n <= 12000 or when the third argument is "cpu" (300 s at n = 40000)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocons_amd as ca                     # noqa: E402
from cocons_amd import workloads as wl     # noqa: E402

g = int(sys.argv[1]) if len(sys.argv) > 1 else 100
delta = float(sys.argv[2]) if len(sys.argv) > 2 else 0.06
n = g * g
locs = wl.grid_locs(g)
X = wl.design_from_locs(locs)["std.covs"]
th = wl.theta_full()
z = wl.synthetic_z(n)
t0 = time.perf_counter()
ci, rp, ent = [], [1], []
cell = {}
for i, (x, y) in enumerate(locs):
    cell.setdefault((int(x / delta), int(y / delta)), []).append(i)
for i, (x, y) in enumerate(locs):
    cx, cy = int(x / delta), int(y / delta)
    cand = np.array(sorted(j for a in (-1, 0, 1) for b in (-1, 0, 1) for j in cell.get((cx + a, cy + b), [])))
    d = np.sqrt(np.sum((locs[cand] - locs[i]) ** 2, axis=1))
    keep = d <= delta
    h = d[keep] / delta
    ci.extend((cand[keep] + 1).tolist())
    ent.extend(((1 - h) ** 4 * (4 * h + 1)).tolist())
    rp.append(len(ci) + 1)
print("pattern: n = %d, nnz = %d (%.1f per row, %.2f %% dense), built in %.1f s" %
      (n, len(ci), len(ci) / n, 100.0 * len(ci) / n / n, time.perf_counter() - t0))


def device_free_bytes():
    """free device memory (hipMemGetInfo through ctypes): the handle's footprint is the drop across its creation"""
    import ctypes
    from cocons_amd.shard import _hip_runtime
    hip = _hip_runtime()
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
    return free.value


ca.CoconsFit(locs[:300], X[:300], z[:300], wl.SMOOTH_LIMITS).neg2loglik_core(th)   # library + context are up
free0 = device_free_bytes()
fit = ca.CoconsTaperFit(locs, X, z, wl.SMOOTH_LIMITS, np.array(ci, dtype=np.int32), np.array(rp, dtype=np.int32), np.array(ent))
for _ in range(3):
    v, parts = fit.neg2loglik_core(th)
print("device memory of the handle (pattern, data, factorisation buffer): %.3f GB; a dense n x n buffer alone: %.2f GB" %
      ((free0 - device_free_bytes()) / 1e9, 8.0 * n * n / 1e9))
K = 20
t0 = time.perf_counter()
for _ in range(K):
    v, parts = fit.neg2loglik_core(th)
dt = (time.perf_counter() - t0) / K
print("taper objective: %.3f ms per evaluation (%.1f evals/s), value %.6f" % (1e3 * dt, 1 / dt, v))
nb = 33                                           # one finite-difference gradient of a 16-parameter model
tls = []
for i in range(nb):
    t2 = {k: np.array(v, dtype=float) for k, v in th.items()}
    t2["std.dev"][0] += 1e-3 * i
    tls.append(t2)
fit.neg2loglik_batch_core(tls[:4])
t0 = time.perf_counter()
vals, st = fit.neg2loglik_batch_core(tls)
dtb = time.perf_counter() - t0
print("batch of %d independent evaluations: %.1f evals/s (all ok: %s)" % (nb, nb / dtb, bool(np.all(st == 0))))
# context: a sparse direct factorisation of the same matrix on this host's CPU (scipy / SuperLU, one thread's worth of
# work; spam's supernodal Cholesky is not available here and would be roughly 2x cheaper than an LU)
if n > 12000 and not (len(sys.argv) > 3 and sys.argv[3] == "cpu") or (len(sys.argv) > 3 and sys.argv[3] == "nocpu"):
    sys.exit(0)
try:
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    vals = np.array(ent) * ca.cov_rns_taper(th, locs, X, np.array(ci, dtype=np.int32), np.array(rp, dtype=np.int32), wl.SMOOTH_LIMITS)
    S = sp.csr_matrix((vals, np.array(ci) - 1, np.array(rp) - 1), shape=(n, n)).tocsc()
    t0 = time.perf_counter()
    lu = spl.splu(S, permc_spec="MMD_AT_PLUS_A", options=dict(SymmetricMode=True))
    resid = z[:, 0] - X @ th["mean"] if z.ndim == 2 else z - X @ th["mean"]
    q = float(resid @ lu.solve(resid))
    dtc = time.perf_counter() - t0
    ld = float(np.sum(np.log(np.abs(lu.U.diagonal()))))
    vc = n * np.log(2 * np.pi) + ld + q
    print("CPU sparse LU of the same matrix: %.1f ms; value %.6f (rel. diff %.1e)" % (1e3 * dtc, vc, abs(vc - v) / abs(v)))
except Exception as e:          # noqa: BLE001
    print("CPU sparse comparison skipped:", e)
