#!/usr/bin/env python3
"""The persistent launch of the dependency-driven schedule (cocons::dag_kernel) replayed ALONE at n = 10^4
(cocons_debug_dag_replay): the program the counter passes of `tools/gpu_run.sh pmc_dag` run under rocprofv3 --pmc, where the real
launch cannot run (kernels are serialised there and it waits for the diagonal-block engine on another stream).

  python3 tools/dag_replay.py [--n 10000] [--reps 3] [--warm 1]

Prints one JSON line: duration of the replayed launch (HIP events), its update flops, TFLOP/s, and the check -- the panels
the replay formed against the plain-schedule factor of the same matrix."""
import argparse
import ctypes
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--warm", type=int, default=1)
    a = ap.parse_args()
    import cocons_amd as ca
    from cocons_amd import _lib, workloads as wl
    from cocons_amd.host import theta_table, _p
    g = int(round(math.sqrt(a.n)))
    locs = wl.grid_locs(g)
    X = wl.design_from_locs(locs)["std.covs"]
    th = wl.theta_full()
    fit = ca.CoconsFit(locs, X, wl.synthetic_z(g * g), wl.SMOOTH_LIMITS)
    L = _lib.load()
    T = theta_table(th)
    mean = np.ascontiguousarray(np.asarray(th["mean"], dtype=np.float64))
    out = np.zeros(5)
    for _ in range(a.warm):
        _lib.check(L.cocons_debug_dag_replay(fit._h, _p(T), _p(mean), 1, _p(out)), "cocons_debug_dag_replay")
    _lib.check(L.cocons_debug_dag_replay(fit._h, _p(T), _p(mean), a.reps, _p(out)), "cocons_debug_dag_replay")
    res = {"n": g * g, "kernel": "cocons::dag_kernel replayed alone (engine outputs from a plain-schedule factorisation)",
           "launch_ms": round(out[0], 4), "update_flops": out[1], "tflops": round(out[1] / (out[0] * 1e-3) / 1e12, 3),
           "panels_vs_plain_factor_rel": out[2], "tasks": int(out[3]), "steps": int(out[4]), "reps": a.reps}
    print(json.dumps(res))
    assert out[2] < 1e-9, "the replayed launch formed other panels than the plain factorisation"
    fit.close()


if __name__ == "__main__":
    main()
