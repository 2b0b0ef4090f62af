import sys, numpy as np, ctypes
sys.path.insert(0,'/root/repo')
from cocons_amd import _lib
lib=_lib.load()
n=100
rng=np.random.default_rng(0)
B=rng.standard_normal((n,n)); A=np.asfortranarray(B@B.T+n*np.eye(n))
dp=ctypes.POINTER(ctypes.c_double); ld=ctypes.c_double()
print("calling", flush=True)
rc=lib.cocons_chol_solve(n, A.ctypes.data_as(dp), 0, None, None, None, ctypes.byref(ld))
print(rc, ld.value, np.sum(np.log(np.diag(np.linalg.cholesky(A)))), flush=True)
