#!/usr/bin/env python3
"""Several handles of ONE process evaluating at the same time (optimParallel's workers in threads): throughput, and what the
engine / DAG schedule of each handle made of the competition (time-outs, last abort code).
    python tools/diag/inflight_probe.py [n_side=100] [handles=2] [evaluations per handle=40]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cocons_amd as ca                     # noqa: E402
from cocons_amd import workloads as wl     # noqa: E402

g = int(sys.argv[1]) if len(sys.argv) > 1 else 100
H = int(sys.argv[2]) if len(sys.argv) > 2 else 2
per = int(sys.argv[3]) if len(sys.argv) > 3 else 40
locs = wl.grid_locs(g)
X = wl.design_from_locs(locs)["std.covs"]
z = wl.synthetic_z(g * g)
th = wl.theta_full()
fits = [ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS) for _ in range(H)]
ref = fits[0].neg2loglik_core(th)[0]
for f in fits:
    assert abs(f.neg2loglik_core(th)[0] - ref) <= 1e-11 * abs(ref)
t0 = time.perf_counter()
for _ in range(per):
    fits[0].neg2loglik_core(th)
seq = per / (time.perf_counter() - t0)
bad = [0] * H


def worker(i):
    for _ in range(per):
        v = fits[i].neg2loglik_core(th)[0]
        if abs(v - ref) > 1e-11 * abs(ref):
            bad[i] += 1


ts = [threading.Thread(target=worker, args=(i,)) for i in range(H)]
t0 = time.perf_counter()
for t in ts:
    t.start()
for t in ts:
    t.join()
dt = time.perf_counter() - t0
print("n = %d: sequential %.1f evals/s; %d handles in flight: %.1f evals/s; wrong values %s" % (g * g, seq, H, H * per / dt, bad))
for i, f in enumerate(fits):
    print("  handle %d: %s" % (i, f.engine_state()))
    f.close()
