"""Times and hashes the 64x64 factorisation with every factor-phase variant (EKFVIO_POTRF_FV) on the same SPD block."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF
g = TightlyCoupledEKF(max_features=4, hooks=True)
for fv in (sys.argv[1:] or ["0", "8", "10", "12", "13"]):
    os.environ["EKFVIO_POTRF_FV"] = fv
    best = None
    for rep in range(6):
        st = (C.c_int64 * 80)()
        assert g.lib.ekfvio_test_potrf_stamps(g.h, st) == 0
        v = list(st)
        tot = v[11] - v[0]
        best = tot if best is None else min(best, tot)
    fac = [v[2 + 2 * p] - v[1 + 2 * p] for p in range(4)]
    print("FV %2s: total %6d ticks (best of 6)  factor phases %s  trails %s  hash L %016x inv %016x" % (
        fv, best, fac, [v[3 + 2 * p] - v[2 + 2 * p] for p in range(3)], v[12] & (2**64 - 1), v[13] & (2**64 - 1)))
