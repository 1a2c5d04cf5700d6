"""In-kernel stamps (s_memtime cycles) of chol_persist_la_kernel at N >= 512: the chain's rounds, the sub-diagonal row worker's near visits, one far owner."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF
from ekf_vio_amd.sim import Scenario
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sc = Scenario(N, seed=0)
g = TightlyCoupledEKF(max_features=N)
g.addNewFeatures(sc.initial_features())
g.lib.ekfvio_test_sweep_stamps(g.h, 1, None)
for z, R, p in sc.frames(10):
    g.process(sc.dt)
    g.updateWithFeaturePositions(z, R, p)
st = (C.c_int64 * 1024)()
g.lib.ekfvio_test_sweep_stamps(g.h, 1, st)
v = list(st)
mb = (2 * N + 63) // 64
names = ["wait", "loads", "product, solve, product", "factorisation", "stores, publish"]
t0 = v[0]
for l in range(mb - 1):
    e = v[8 * l:8 * l + 6]
    nxt = v[8 * (l + 1)] if l + 2 < mb else e[5]
    print("chain round %2d at %7d: " % (l, e[0] - t0) + "  ".join("%s %5d" % (names[k], e[k + 1] - e[k]) for k in range(5)) + "   | %6d cycles" % (e[5] - e[0]))
print("chain total %d cycles = %.1f us at 2.4 GHz" % (v[8 * (mb - 2) + 5] - t0, (v[8 * (mb - 2) + 5] - t0) / 2400.0))
rn = ["wait", "stored product + L_ll", "preload, tiles", "solve", "store, product, store", "publish"]
for l in range(mb - 2):
    e = v[300 + 8 * l:300 + 8 * l + 7]
    if e[0] and e[6]:
        print("row worker of tile (%d,%d), round %2d: " % (l + 2, l + 1, l) + "  ".join("%s %5d" % (rn[k], e[k + 1] - e[k]) for k in range(6)) + "   | %6d" % (e[6] - e[0]))
for q in range(32):
    e = v[600 + 8 * q:600 + 8 * q + 3]
    if e[0] and e[2]:
        print("far visit (round %% 32 = %2d): wait %5d  work %5d" % (q, e[1] - e[0], e[2] - e[1]))
