#!/usr/bin/env python3
"""Print the sustained fp64 MFMA rate of this GPU (cocons_mfma_f64_probe)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cocons_amd import _lib

L = _lib.load_probes()          # libcocons_hip_probes.so: the probes are not in the product library
for bpc in (1, 2, 4, 8):
    t = ctypes.c_double()
    _lib.check(L.cocons_mfma_f64_probe(bpc, ctypes.byref(t)), "probe")
    print("v_mfma_f64_16x16x4_f64 back-to-back, %d block(s) of 4 waves per CU: %.2f TFLOP/s" % (bpc, t.value))
for bpc in (1, 2, 4, 8):
    t = ctypes.c_double()
    _lib.check(L.cocons_vfma_f64_probe(bpc, ctypes.byref(t)), "probe")
    print("v_fma_f64 independent chains (16 per lane), %d block(s) of 4 waves per CU: %.2f TFLOP/s" % (bpc, t.value))
