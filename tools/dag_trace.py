#!/usr/bin/env python3
"""Where the time of the dependency-driven factorisation (dag_kernel) goes: per-task stamps of one evaluation
(cocons_debug_dag_trace) summarised per step and per task kind.

    python tools/dag_trace.py [--n 10000] [--lead 1600]
"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--lead", type=int, default=1600)
    ap.add_argument("--every", type=int, default=4, help="print every k-th step")
    ap.add_argument("--chain", type=int, default=-1, help="also print the stamps of the chain's panel tasks of this step")
    a = ap.parse_args()
    import cocons_amd as ca
    from cocons_amd import _lib, workloads as wl
    L = _lib.load()
    tune = lambda k, v: _lib.check(L.cocons_debug_tune(k.encode(), int(v)), "tune")
    tune("dag", 1); tune("dag_lead", a.lead); tune("dag_trace", 1)
    g = int(round(a.n ** 0.5))
    locs = wl.grid_locs(g)
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full()
    z = wl.synthetic_z(g * g)
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
    for _ in range(3):
        fit.neg2loglik_core(th)
    ns = ctypes.c_int(0)
    nt = L.cocons_debug_dag_trace(fit._h, ctypes.byref(ns), None, None, None)
    assert nt > 0, _lib.last_error()
    steps = np.zeros((ns.value, 16), dtype=np.int32)     # DagStepHost: 16 words
    st = np.zeros((nt, 4), dtype=np.uint64)
    eng = np.zeros((8 * (2 * ns.value + 8),), dtype=np.uint64)
    ntile = (g * g + 127) // 128
    eng = np.zeros((8 * (ntile + 2),), dtype=np.uint64)
    rc = L.cocons_debug_dag_trace(fit._h, ctypes.byref(ns), steps.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                  st.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)),
                                  eng.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)))
    assert rc == nt, _lib.last_error()
    t0 = float(st[:, 0].min())
    T = (st.astype(np.float64) - t0) * 0.01                    # microseconds
    E = (eng.astype(np.float64).reshape(-1, 8) - t0) * 0.01
    total = T[:, 3].max()
    print("n = %d: %d tasks in %d steps, kernel span %.1f us; slot-time busy %.1f %% of 2040 slots"
          % (g * g, nt, ns.value, total, 100 * np.sum(T[:, 3] - T[:, 0]) / (2040 * total)))
    print("per step: tasks; [start end] of the step; far tiles: mean wait / product / epilogue; then the CHAIN of the step (us, absolute):")
    print("  diagR = the 10 diagonal-block tiles have their inputs, diagE = last of them stored (product time in brackets);")
    print("  engine: tile t factored +inverse-> out[t]; xr; tile t+1 factored +inverse-> out[t+1];  T3c = T3 of the first four strips done")
    kinds_tot = {}
    prev_t3c = 0.0
    for s in range(ns.value):
        base, near, tpos, nT = [int(v) for v in steps[s].view(np.uint32)[:4]]
        H, W, tj0, k0, K, nstrip, two, need, nd_next, split = [int(v) for v in steps[s][4:14]]
        p2, p3 = [int(v) for v in steps[s].view(np.uint32)[14:16]]
        nxt = int(steps[s + 1].view(np.uint32)[0]) if s + 1 < ns.value else nt
        rows = T[base:nxt]
        q = np.arange(nxt - base)
        # the list of a step: tiles | T1, early halves | tiles | T2 | tiles | T3 | tiles
        per = 2 * nstrip
        nA, perBC = per + nd_next, (per if two else 0)
        inA = (q >= tpos) & (q < tpos + nA)
        inB = (q >= p2) & (q < p2 + perBC)
        inC = (q >= p3) & (q < p3 + perBC)
        isT = inA | inB | inC
        early = inA & (q - tpos >= per)                          # early halves of the next step's diagonal-block tiles
        stage = np.where(inA & ~early, 0, np.where(inB, 1, np.where(inC, 2, -1)))
        u2 = np.where(inA, q - tpos, np.where(inB, q - p2, q - p3))
        strip = np.where(isT & ~early, u2 // 2, -1)
        isnear = (~isT) & (q < near)
        isfar = (~isT) & ~isnear
        # diagonal-block tiles: column jl < 4, row offset jl + r < 4
        isdiag = np.zeros_like(isnear)
        off = 0
        for jl in range(min(W, 4)):
            cnt = H - jl
            for r in range(min(cnt, 4 - jl)):
                isdiag[off + r] = True
            off += cnt

        def stats(m):
            if not m.any():
                return (0.0, 0.0, 0.0)
            r = rows[m]
            return (np.mean(r[:, 1] - r[:, 0]), np.mean(r[:, 2] - r[:, 1]), np.mean(r[:, 3] - r[:, 2]))
        sf = stats(isfar)
        for name, m in (("diag", isdiag), ("near", isnear & ~isdiag), ("far", isfar), ("T1", stage == 0), ("T2", stage == 1), ("T3", stage == 2),
                        ("early", early)):
            if m.any():
                r = rows[m]
                d = kinds_tot.setdefault(name, [0, 0.0, 0.0, 0.0])
                d[0] += int(m.sum()); d[1] += float(np.sum(r[:, 1] - r[:, 0])); d[2] += float(np.sum(r[:, 2] - r[:, 1])); d[3] += float(np.sum(r[:, 3] - r[:, 2]))
        if s % a.every == 0 or s >= ns.value - 6:
            dg = rows[isdiag]
            e = E[(tj0 // 2) // 2]
            last_stage = 2 if two else 0
            m3 = (stage == last_stage) & (strip < 4)
            t3c = rows[m3, 3].max() if m3.any() else float("nan")
            print("%3d %6d [%7.1f %7.1f] far %4.1f/%4.1f/%4.1f | diagR %7.1f diagE %7.1f (%4.1f) | eng %7.1f ->%7.1f; xr %7.1f; %7.1f ->%7.1f | T3c %7.1f | period %6.1f"
                  % (s, nxt - base, rows[:, 0].min(), rows[:, 3].max(), sf[0], sf[1], sf[2], dg[:, 1].max(), dg[:, 3].max(),
                     np.mean(dg[:, 3] - dg[:, 1]), e[1], e[2], e[4], e[6], e[7], t3c, t3c - prev_t3c))
            prev_t3c = t3c
        else:
            last_stage = 2 if two else 0
            m3 = (stage == last_stage) & (strip < 4)
            prev_t3c = rows[m3, 3].max() if m3.any() else prev_t3c
    if a.chain >= 0:
        # the panel tasks of the first four strips of one step (the rows of the next diagonal block): drawn / inputs / product / stored
        s_ = a.chain
        base, near, tpos, nT = [int(v) for v in steps[s_].view(np.uint32)[:4]]
        H, W, tj0, k0, K, nstrip, two, need, nd_next, split = [int(v) for v in steps[s_][4:14]]
        p2, p3 = [int(v) for v in steps[s_].view(np.uint32)[14:16]]
        e = E[(tj0 // 2) // 2]
        print("chain of step %d: engine out[t] %.1f xr %.1f out[t+1] %.1f" % (s_, e[2], e[4], e[7]))
        per = 2 * nstrip
        for stage in range(3 if two else 1):
            for strip in range(min(4, nstrip)):
                for h in range(2):
                    L = base + (tpos, p2, p3)[stage] + 2 * strip + h
                    print("   T%d strip %d h %d (task %d): drawn %8.1f inputs %8.1f product %8.1f stored %8.1f" %
                          (stage + 1, strip, h, L, T[L, 0], T[L, 1], T[L, 2], T[L, 3]))
        for dd in range(nd_next):
            L = base + tpos + per + dd
            print("   early half %2d (task %d): drawn %8.1f inputs %8.1f product %8.1f stored %8.1f" % (dd, L, T[L, 0], T[L, 1], T[L, 2], T[L, 3]))
    print("totals per kind: count, slot-ms waiting for inputs, in the product (+ wait for the previous C version), in the epilogue")
    for k, d in kinds_tot.items():
        print("  %-5s %7d  %9.2f %9.2f %9.2f   mean product %.1f us" % (k, d[0], d[1] * 1e-3, d[2] * 1e-3, d[3] * 1e-3, d[2] / max(1, d[0])))
    fit.close()


if __name__ == "__main__":
    main()
