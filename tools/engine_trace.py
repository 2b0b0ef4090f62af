#!/usr/bin/env python3
"""The diagonal-block engine's eight stamps per 256-column block (100 MHz clock), under the dependency-driven schedule for every
step (the stamps exist for DAG blocks only), one-workgroup engine against the pair:

  python3 tools/engine_trace.py [--n 4096] [--pair 1] [--every 1]

per block: first tile -- in[t] seen .. factored .. out[t]; second tile -- in[t+1] seen, xr[t], factored, out[t+1] (all relative to
"in[t] seen"), then the gap to the next block's "in[t] seen"."""
import argparse
import ctypes
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--pair", type=int, default=1)
    ap.add_argument("--min-tiles", type=int, default=0)
    ap.add_argument("--every", type=int, default=1)
    a = ap.parse_args()
    import cocons_amd as ca
    from cocons_amd import _lib, workloads as wl
    L = _lib.load()
    for k, v in (("dag", 1), ("dag_min_tiles", a.min_tiles), ("dag_trace", 1), ("engine_pair", a.pair)):
        _lib.check(L.cocons_debug_tune(k.encode(), v), "tune")
    g = int(round(math.sqrt(a.n)))
    locs = wl.grid_locs(g)
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full()
    fit = ca.CoconsFit(locs, X, wl.synthetic_z(g * g), wl.SMOOTH_LIMITS)
    for _ in range(3):
        fit.neg2loglik_core(th)
    ns = ctypes.c_int(0)
    nt_tasks = L.cocons_debug_dag_trace(fit._h, ctypes.byref(ns), None, None, None)
    steps = np.zeros((ns.value, 16), dtype=np.int32)
    stamps = np.zeros((nt_tasks, 4), dtype=np.uint64)
    nt = (fit.n + 127) // 128 + 2
    eng = np.zeros((nt + 2, 8), dtype=np.uint64)
    L.cocons_debug_dag_trace(fit._h, ctypes.byref(ns), steps.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                             stamps.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), eng.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)))
    print("n = %d pair = %d: %d steps; engine state %s" % (fit.n, a.pair, ns.value, fit.engine_state()))
    print("%5s | %8s %8s | %8s %8s %8s %8s | %8s %8s" % ("block", "factored", "out[t]", "in[t+1]", "xr[t]", "factored", "out[t+1]",
                                                        "block us", "to next"))
    rows = []
    for p in range(1, nt // 2 + 1):
        e = eng[p].astype(np.int64)
        if e[0] == 0 or e[7] == 0:
            continue
        nxt = eng[p + 1].astype(np.int64)
        rel = (e - e[0]) * 0.01
        gap = (nxt[0] - e[7]) * 0.01 if nxt[0] else float("nan")
        rows.append((rel[7], gap, rel[1], rel[6] - rel[1]))
        if (p - 1) % a.every == 0:
            print("%5d | %8.1f %8.1f | %8.1f %8.1f %8.1f %8.1f | %8.1f %8.1f" % (p, rel[1], rel[2], rel[3], rel[4], rel[6], rel[7], rel[7], gap))
    r = np.array(rows)
    half = len(r) // 2
    print("median over the last half of the blocks: block %.1f us (first tile %.1f, first tile factored -> second factored %.1f), gap to next %.1f"
          % (np.median(r[half:, 0]), np.median(r[half:, 2]), np.median(r[half:, 3]), np.nanmedian(r[half:, 1])))


if __name__ == "__main__":
    main()
