#!/usr/bin/env python3
"""Assembly BESIDE the factorisation (round 6): what do the covariance assembly (fp64 vector work) and the
factorisation (fp64 matrix work) cost each other when they share the chip?  Two handles at n = 10^4: handle A evaluates -2 loglik
in a loop (assembly + factorisation, the bench line's step), handle B runs ONLY the assembly in a loop (cocons_debug_assembly_loop).
Each alone, then both at once from two host threads.  If the two kinds of work used separate resources, both rates would stay
near their solo values (sum of the relative rates ~ 2); if they compete for one resource the sum stays ~ 1.
    python tools/diag/overlap_probe.py [n = 10000]"""
import ctypes
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cocons_amd as ca                     # noqa: E402
from cocons_amd import _lib, workloads as wl     # noqa: E402
from cocons_amd.host import theta_table          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
g = int(round(n ** 0.5))
locs = wl.grid_locs(g)
X = wl.design_from_locs(locs)["std.covs"]
z = wl.synthetic_z(g * g)
th = wl.theta_full()
T = theta_table(th)
L = _lib.load()
A = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
B = ca.CoconsFit(locs, X, z, wl.SMOOTH_LIMITS)
c_dp = ctypes.POINTER(ctypes.c_double)


def asm_loop(reps):
    out = np.zeros(1)
    _lib.check(L.cocons_debug_assembly_loop(B._h, T.ctypes.data_as(c_dp), reps, out.ctypes.data_as(c_dp)), "assembly_loop")
    return out[0]


def evals(k):
    t0 = time.perf_counter()
    for _ in range(k):
        A.neg2loglik_core(th)
    return k / (time.perf_counter() - t0)


evals(5)
asm_loop(5)
solo_eval = max(evals(40) for _ in range(2))
solo_asm = min(asm_loop(40) for _ in range(2))
st = A.profile_stages(th, reps=3)
print("n = %d" % (g * g))
print("solo: %.1f evaluations/s (%.3f ms each: assembly %.3f + factorisation %.3f), assembly loop %.3f ms per assembly"
      % (solo_eval, 1e3 / solo_eval, st["assembly_ms"], st["cholesky_ms"], solo_asm))
res = {}
stop = threading.Event()
asm_ms = []


def side():
    while not stop.is_set():
        asm_ms.append(asm_loop(10))


tb = threading.Thread(target=side)
tb.start()
time.sleep(0.05)
res["eval"] = evals(120)
stop.set()
tb.join()
both_asm = float(np.median(asm_ms))
ra, rb = res["eval"] / solo_eval, solo_asm / both_asm
print("together: %.1f evaluations/s (x%.3f of solo), assembly loop %.3f ms per assembly (x%.3f of its solo rate; %d samples)"
      % (res["eval"], ra, both_asm, rb, len(asm_ms)))
print("sum of the relative rates: %.3f   (1 = the two loops share one resource, 2 = they do not compete)" % (ra + rb))
# what that would mean for ONE evaluation whose assembly were hidden under its own factorisation
chol, asm = st["cholesky_ms"], st["assembly_ms"]
# time in which the chip does one factorisation plus one assembly's worth of side work, at the measured mixed rates
t_mixed = 1e3 / res["eval"]                  # per evaluation of A (its own assembly + factorisation), B's work riding along
side_per_eval = t_mixed / both_asm           # assemblies B completes meanwhile
print("per evaluation of A (%.3f ms) the chip also completed %.2f extra assemblies: throughput of assembly + factorisation work "
      "together = %.3f x sequential" % (t_mixed, side_per_eval, (1.0 + side_per_eval * asm / (asm + chol)) * (1e3 / solo_eval) / t_mixed))
print("engine states", A.engine_state(), B.engine_state())
A.close()
B.close()
