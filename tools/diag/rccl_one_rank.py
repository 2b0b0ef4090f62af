import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cocons_amd import _lib, workloads as wl
from cocons_amd.shard import ShardedFit, MultiFit
g = int(sys.argv[1]) if len(sys.argv) > 1 else 20
locs = wl.grid_locs(g); X = wl.design_from_locs(locs)["std.covs"]; th = wl.theta_full()
z = wl.synthetic_z(g * g)
fit = ShardedFit(locs, X, z, wl.SMOOTH_LIMITS, device=0)
L = _lib.load()
idb = (ctypes.c_ubyte * 128)()
print("uid rc", L.cocons_comm_unique_id(ctypes.cast(idb, ctypes.c_void_p)), _lib.last_error())
rc = L.cocons_fit_comm_init(fit._h, 1, 0, ctypes.cast(idb, ctypes.c_void_p))
print("init rc", rc, _lib.last_error())
if rc == 0:
    print(fit.neg2loglik_core(th))
try:
    mf = MultiFit(locs, X, z, wl.SMOOTH_LIMITS, devices=[0])
    print("multi", mf.neg2loglik_core(th))
except Exception as e:
    print("multi failed", e)
