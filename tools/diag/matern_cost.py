"""Cost of the device Matern correlation per argument range: N points at one u (nu jittered in [0.5, 2.5]) per
launch of matern_points_kernel; run under `rocprofv3 --kernel-trace` and read the durations (tools/diag/matern_cost_read.py)."""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cocons_amd import _lib

L = _lib.load()
N = 1 << 22
rng = np.random.default_rng(1)
nu = rng.uniform(0.5, 2.5, N)
out = np.empty(N)
US = [0.05, 0.5, 1.0, 1.9, 2.1, 3.0, 4.0, 6.0, 10.0, 19.0, 21.0, 40.0, 80.0]
for u0 in US:
    u = u0 * rng.uniform(0.98, 1.02, N)
    _lib.check(L.cocons_debug_matern(N, nu.ctypes.data_as(_lib.c_dp), u.ctypes.data_as(_lib.c_dp),
                                     out.ctypes.data_as(_lib.c_dp)), "cocons_debug_matern")
print("order:", US)
