#!/usr/bin/env python3
"""Sequential against batched throughput (cocons_neg2loglik_batch, 33 points = one central-difference gradient at P = 16)
at n = 4096 and n = 10 000; the environment selects slots / engine use / hardware queues (GPU_MAX_HW_QUEUES)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cocons_amd as ca
from cocons_amd import workloads as wl

tag = " ".join("%s=%s" % kv for kv in sorted(os.environ.items()) if kv[0].startswith(("COCONS_", "GPU_MAX")))
for g in (64, 100):
    locs = wl.grid_locs(g)
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full()
    fit = ca.CoconsFit(locs, X, wl.synthetic_z(g * g), wl.SMOOTH_LIMITS)
    ths = []
    for i in range(33):
        t = {k: np.array(v, dtype=np.float64) for k, v in th.items()}
        t["std.dev"][0] += 1.22e-4 * (i + 1)
        ths.append(t)
    for t in ths[:5]:
        fit.neg2loglik_core(t)
    t0 = time.perf_counter()
    for t in ths:
        fit.neg2loglik_core(t)
    dts = time.perf_counter() - t0
    fit.neg2loglik_batch_core(ths[:6])
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        v, s = fit.neg2loglik_batch_core(ths)
        best = min(best, time.perf_counter() - t0)
    from cocons_amd import _lib
    eo = np.zeros(2)
    _lib.load().cocons_debug_host_enqueue(fit._h, eo.ctypes.data_as(_lib.c_dp))
    print("[%s] n=%d: sequential %.1f evals/s, batch of 33 %.1f evals/s (x%.2f), all ok %s, engine %s, host enqueue %.0f us per evaluation (%d)"
          % (tag, g * g, 33 / dts, 33 / best, dts / best, bool((s == 0).all()), fit.engine_state(), eo[0], int(eo[1])))
    fit.close()
