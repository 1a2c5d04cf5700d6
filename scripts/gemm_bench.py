"""Times the fp32 MFMA GEMM at the filter's shapes (run on the GPU box)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF  # noqa: E402

g = TightlyCoupledEKF(max_features=4, hooks=True)
shapes = [("joseph N=256", 1, 0, 790, 790, 512), ("gain   N=256", 1, 1, 790, 512, 512), ("gainNN N=256", 0, 1, 790, 512, 512),
          ("joseph N=64 ", 1, 0, 214, 214, 128), ("joseph N=1024", 1, 0, 3094, 3094, 2048), ("square 4096", 1, 0, 4096, 4096, 4096)]
variants = [int(v) for v in (sys.argv[1:] or ["0"])]
for name, tb, lo, M, N, K in shapes:
    for v in variants:
        us = C.c_double(0)
        reps = 200 if M < 1000 else (20 if M < 4000 else 5)
        rc = g.lib.ekfvio_test_gemm_bench(g.h, tb, lo, M, N, K, reps, v, C.byref(us))
        fl = 2.0 * M * N * K * (0.5 if lo else 1.0)
        print("%-14s variant %d  rc %d  %8.2f us  %7.2f TFLOP/s (%4.1f%% of 157.3)" % (name, v, rc, us.value, fl / us.value / 1e6, fl / us.value / 1e6 / 1.573), flush=True)
