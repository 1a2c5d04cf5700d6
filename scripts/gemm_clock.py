"""One GEMM shape, many reps: for a rocprofv3 --pmc GRBM_GUI_ACTIVE pass (effective clock)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF
g = TightlyCoupledEKF(max_features=4, hooks=True)
M, N, K, reps = [int(x) for x in sys.argv[1:5]]
us = C.c_double(0)
g.lib.ekfvio_test_gemm_bench(g.h, 1, 0, M, N, K, reps, 0, C.byref(us))
print(M, N, K, us.value)
