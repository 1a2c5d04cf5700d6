"""Shape sweep of the fp32 MFMA GEMM: separates fixed launch/prologue/epilogue cost from the
per-K-tile cost, for one workgroup and for a full grid (run on the GPU box)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF  # noqa: E402

g = TightlyCoupledEKF(max_features=4, hooks=True)
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for M, N in [(64, 64), (128, 128), (790, 790), (1024, 1024), (2048, 2048)]:
    row = []
    for K in (64, 128, 512, 2048):
        us = C.c_double(0)
        g.lib.ekfvio_test_gemm_bench(g.h, 1, 0, M, N, K, 100, variant, C.byref(us))
        row.append(us.value)
    per_tile = (row[3] - row[2]) / ((2048 - 512) / 32)
    wgs = ((M + 63) // 64) * ((N + 63) // 64)
    print("M=N=%4d (%4d WGs)  K=64 %7.2f  K=128 %7.2f  K=512 %7.2f  K=2048 %8.2f us | per K-tile %.3f us = %.0f MFMA-cycles-equivalents at 2.4GHz"
          % (M, wgs, row[0], row[1], row[2], row[3], per_tile, per_tile * 2400), flush=True)
