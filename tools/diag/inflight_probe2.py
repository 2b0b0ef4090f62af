#!/usr/bin/env python3
"""Throughput with T host threads, each its own handle, evaluating concurrently (what separate worker processes or a
threaded batch would do) against the single-thread batch entry, at n = 4096 and 10 000."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cocons_amd as ca
from cocons_amd import workloads as wl

for g, per in ((64, 60), (100, 24)):
    locs = wl.grid_locs(g)
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full()
    z = wl.synthetic_z(g * g)
    for T in (1, 2, 3, 4):
        fits = [ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS) for _ in range(T)]
        for f in fits:
            f.neg2loglik_core(th)

        def worker(f):
            for _ in range(per):
                f.neg2loglik_core(th)
        ts = [threading.Thread(target=worker, args=(f,)) for f in fits]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        print("n=%d: %d threads x %d evals: %.1f evals/s; engine states %s" % (g * g, T, per, T * per / dt,
              [(f.engine_state()["active"], f.engine_state()["retries"]) for f in fits]), flush=True)
        for f in fits:
            f.close()
