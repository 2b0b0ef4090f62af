#!/usr/bin/env python3
"""Time schedule variants of the factorisation in alternation inside ONE process on ONE device (the only comparison
that means anything on this pool: boxes differ by several percent and so do separate runs on one box).

    python tools/ab_modes.py [--n 10000] [--rounds 7] [--evals 10] "name:key=val,key=val" ...

Each variant is a list of cocons_debug_tune settings applied before its turn; every round times `evals` sequential
evaluations of every variant; the table gives median and minimum ms per evaluation over the rounds.  Values are checked
against the first variant (relative 1e-10)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--evals", type=int, default=10)
    ap.add_argument("--nocheck", action="store_true", help="variants may change the value (timing ablations)")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    import cocons_amd as ca
    from cocons_amd import _lib, workloads as wl
    L = _lib.load()
    g = int(round(a.n ** 0.5))
    locs = wl.grid_locs(g)
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full()
    z = wl.synthetic_z(g * g)
    fit = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
    variants = []
    for v in a.variants:
        name, _, kv = v.partition(":")
        sets = [(k.split("=")[0], int(k.split("=")[1])) for k in kv.split(",") if k]
        variants.append((name, sets))
    defaults = {"engine": 1, "upd_dynamic": 1, "upd_waves": 8, "w8_max_tiles": 3500, "c_wt": 0, "dag": 1, "dag_lead": 1600, "dag_lead2": 600,
                "dag_lead3": 1800, "dag_min_tiles": 2000, "dag_split": 1, "dag_xcd": 1, "dag_order": 1, "dag_bw": 16, "dag_bh": 16, "engine_pair": 1, "panel_fused": 1, "potrf_follow": 1, "panel_follow": 1, "panel_diag": 1, "panel_split": 32, "engine_block0": 1}

    def apply(sets):
        for k, val in defaults.items():
            rc = L.cocons_debug_tune(k.encode(), val)
            if rc != 0 and not os.environ.get("COCONS_HIP_LIB"):      # (an older build alternated on the box lacks the newer switches)
                _lib.check(rc, "tune")
        for k, val in sets:
            _lib.check(L.cocons_debug_tune(k.encode(), val), "tune")

    if a.nocheck:                                     # ablated kernels leave garbage: a failed factorisation is expected
        core = fit.neg2loglik_core

        def tolerant(theta):
            try:
                return core(theta)
            except Exception:                         # noqa: BLE001
                return (float("nan"), None)
        fit.neg2loglik_core = tolerant
    times = {name: [] for name, _ in variants}
    stages = {name: [] for name, _ in variants}
    ref = None
    for name, sets in variants:                       # warm-up + value check
        apply(sets)
        v = fit.neg2loglik_core(th)[0]
        fit.neg2loglik_core(th)
        if ref is None:
            ref = v
        assert a.nocheck or abs(v - ref) <= 1e-10 * abs(ref), (name, v, ref)
    for r in range(a.rounds):
        for name, sets in variants:
            apply(sets)
            fit.neg2loglik_core(th)
            t0 = time.perf_counter()
            for _ in range(a.evals):
                fit.neg2loglik_core(th)
            times[name].append((time.perf_counter() - t0) / a.evals * 1e3)
            try:
                st = fit.profile_stages(th, reps=2)
                stages[name].append((st["assembly_ms"], st["cholesky_ms"], st["update_sum_ms"]))
            except Exception:                         # noqa: BLE001  (--nocheck: the ablated factorisation fails)
                stages[name].append((float("nan"),) * 3)
    print("n = %d, %d rounds x %d evaluations, engine retries %d" % (g * g, a.rounds, a.evals, fit.engine_state()["retries"]))
    base = np.median(times[variants[0][0]])
    for name, _ in variants:
        t = np.array(times[name])
        s = np.median(np.array(stages[name]), axis=0)
        print("%-14s median %7.3f ms  min %7.3f  (%.1f evals/s, %+5.1f%% vs first)   asm %.3f chol %.3f updsum %.3f"
              % (name, np.median(t), t.min(), 1e3 / np.median(t), (base / np.median(t) - 1) * 100, s[0], s[1], s[2]))


if __name__ == "__main__":
    main()
