import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "cocons_amd", "csrc", "libcocons_hip_probes.so"))
L.cocons_corun_probe.argtypes = [ctypes.c_int] * 4 + [ctypes.POINTER(ctypes.c_double)]
out = np.zeros(4)
for bm, bv in ((4, 4), (4, 2), (2, 4), (6, 2), (8, 0), (0, 8)):
    im = 200000 if bm else 0
    iv = 400000 if bv else 0
    # scale iterations so that both take about the same time
    rc = L.cocons_corun_probe(max(bm, 1) if bm else 1, max(bv, 1) if bv else 1, im if bm else 10, iv if bv else 10,
                              out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    print("waves/SIMD mfma=%d vfma=%d: MFMA %.1f TF (%.1f ms)  FMA %.1f TF (%.1f ms)  sum %.1f" %
          (bm, bv, out[0] if bm else 0, out[2], out[1] if bv else 0, out[3], (out[0] if bm else 0) + (out[1] if bv else 0)))
