"""Measured ceiling of the bf16 matrix pipe on the igemm engine's instruction mix (tools/gpu/mfma_peak.hip), no memory
traffic: zero vs random split-image operands, 1 and 2 waves per SIMD.  Prints bf16 TFLOP/s, the fp32-equivalent (/6) and
the core clock under load."""
import ctypes
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "gpu", "libmfma_peak.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(here, "gpu", "mfma_peak.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.mfma_peak_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                              ctypes.POINTER(ctypes.c_double)]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
for rnd in (0, 1, 0, 1):
    for bpc in (1, 2):
        tf, mhz = ctypes.c_double(), ctypes.c_double()
        rc = lib.mfma_peak_run(bpc, iters, rnd, ctypes.byref(tf), ctypes.byref(mhz))
        print(f"mfma_peak data={'random-split' if rnd else 'zeros'} waves/SIMD={bpc}: {tf.value:7.1f} bf16 TFLOP/s = "
              f"{tf.value / 6:6.1f} fp32-equivalent (bf16x6), core clock ~{mhz.value:.0f} MHz, rc={rc}", flush=True)
