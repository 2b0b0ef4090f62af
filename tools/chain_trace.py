#!/usr/bin/env python3
"""Where the chain between two diagonal blocks spends its time under the chain layout of the dependency-driven schedule:
the engine's eight stamps per 256-column block and the four stamps of every chain-helper task, for the steps asked for.

  python3 tools/chain_trace.py [--n 10000] [--min-tiles 0] [--steps 14,25,36]"""
import argparse
import ctypes
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tri = [0, 1, 1, 2, 2, 2, 3, 3, 3, 3]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--min-tiles", type=int, default=0)
    ap.add_argument("--steps", default="")
    a = ap.parse_args()
    import cocons_amd as ca
    from cocons_amd import _lib, workloads as wl
    L = _lib.load()
    for k, v in (("dag", 1), ("dag_chain", 1), ("dag_min_tiles", a.min_tiles), ("dag_trace", 1)):
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
    steps = np.zeros((ns.value, 20), dtype=np.int32)
    stamps = np.zeros((nt_tasks, 4), dtype=np.uint64)
    nt = (fit.n + 127) // 128 + 2
    eng = np.zeros((nt + 2, 8), dtype=np.uint64)
    L.cocons_debug_dag_trace(fit._h, ctypes.byref(ns), steps.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                             stamps.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), eng.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)))
    nc = L.cocons_debug_chain_trace(fit._h, None)
    cst = np.zeros((max(nc, 1), 4), dtype=np.uint64)
    L.cocons_debug_chain_trace(fit._h, cst.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)))
    t0 = int(stamps[:, 0][stamps[:, 0] > 0].min())
    us = lambda v: (int(v) - t0) * 0.01
    names = ("in[t] seen", "tile t factored", "out[t]", "in[t+1] seen", "xr[t]", "tile t+1 updated", "factored", "out[t+1]")
    want = [int(x) for x in a.steps.split(",") if x] or list(range(0, ns.value, max(1, ns.value // 6)))
    print("n = %d: %d steps, %d bulk tasks, %d chain tasks; whole launch %.0f us" % (fit.n, ns.value, nt_tasks, nc,
                                                                                   (int(stamps[:, 3].max()) - t0) * 0.01))
    prev_out = None
    per = []
    for s in range(ns.value):
        pair = s + 1                                    # the engine's block of step s: tiles 2 s + 2, 2 s + 3 -> pair index s + 1
        e = eng[pair]
        if e[0] == 0:
            continue
        if prev_out is not None:
            per.append((s, us(e[0]) - prev_out, us(e[7]) - us(e[0]) if e[7] else float("nan")))
        prev_out = us(e[7]) if e[7] else None
    print("per step: (step, us from out[t+1] of the previous block to in[t] seen, engine us in[t] -> out[t+1])")
    print("  " + "  ".join("%d:%.0f/%.0f" % p for p in per))
    for s in want:
        if s >= ns.value:
            continue
        st = steps[s].view(np.uint32)
        ncd, cs, cbase, cnt = int(steps[s][16]), int(steps[s][17]), int(st[18]), int(st[19])
        nd = ncd * (ncd + 1) // 2
        two, nd_next = int(steps[s][10]), int(steps[s][12])
        e = eng[s + 1]
        base = int(e[0]) if e[0] else t0
        rel = lambda v: (int(v) - base) * 0.01 if v else float("nan")
        print("step %d (H = %d): engine " % (s, steps[s][4]) + ", ".join("%s %.1f" % (names[i], rel(e[i])) for i in range(8)))
        groups = [("diag", 0, nd), ("T1", nd, nd + 2 * cs), ("early", nd + 2 * cs, nd + 2 * cs + nd_next),
                  ("T2", nd + 2 * cs + nd_next, nd + 2 * cs + nd_next + (2 * cs if two else 0)),
                  ("T3", nd + 2 * cs + nd_next + (2 * cs if two else 0), cnt)]
        for nm, lo, hi in groups:
            if hi <= lo:
                continue
            c = cst[cbase + lo:cbase + hi]
            print("   %-5s drawn %6.1f..%6.1f  inputs %6.1f..%6.1f  product %6.1f..%6.1f  stored %6.1f..%6.1f   (task: wait %.1f, product %.1f, tail %.1f us on average)"
                  % (nm, rel(c[:, 0].min()), rel(c[:, 0].max()), rel(c[:, 1].min()), rel(c[:, 1].max()), rel(c[:, 2].min()), rel(c[:, 2].max()),
                     rel(c[:, 3].min()), rel(c[:, 3].max()),
                     np.mean((c[:, 1] - c[:, 0]).astype(np.int64)) * 0.01, np.mean((c[:, 2] - c[:, 1]).astype(np.int64)) * 0.01,
                     np.mean((c[:, 3] - c[:, 2]).astype(np.int64)) * 0.01))
    fit.close()


if __name__ == "__main__":
    main()
