import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
from ekf_vio_amd import TightlyCoupledEKF
g = TightlyCoupledEKF(max_features=4, hooks=True)
for M in (3008, 3072, 3094, 3136, 3200, 3520, 3547):
    us = C.c_double(0)
    rc = g.lib.ekfvio_test_gemm_bench(g.h, 1, 0, M, M, 2048, 20, 1, C.byref(us))
    t = ((M + 63) // 64) ** 2
    fl = 2.0 * M * M * 2048
    print("M=N=%d tiles %d = %.2f rounds of 768  %8.1f us  %6.1f TFLOP/s (%.3f)  us per round-up %.1f" % (M, t, t / 768.0, us.value, fl / us.value / 1e6, fl / us.value / 1e6 / 157.3, us.value / -(-t // 768)), flush=True)
