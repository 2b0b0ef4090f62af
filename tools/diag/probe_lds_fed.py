import ctypes, sys, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/cocons_amd") else os.getcwd())
from cocons_amd import _lib
L = _lib.load()
for form, name in ((3, "register-fed (8 distinct operands)"), (2, "LDS-fed")):
  for bpc in (1, 2, 4, 8):
    out = (ctypes.c_double * 4)()
    _lib.check(L.cocons_mfma_f64_probe_ex(bpc, 16, form, 4000, 0, 3, out), "p")
    print(name + " 4x4x4 loop, %d blocks/CU: %.2f TFLOP/s, clock %.3f GHz, %.2f cycles per MFMA per wave, burst %.3f ms" % (bpc, out[0], out[1], out[2], out[3]))
