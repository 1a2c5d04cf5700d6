"""In-kernel phase stamps (s_memtime cycles) of the chain workgroup of chol_step_kernel during a filter update."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF
from ekf_vio_amd.sim import Scenario
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sc = Scenario(N, seed=0)
g = TightlyCoupledEKF(max_features=N, hooks=True)
g.addNewFeatures(sc.initial_features())
g.lib.ekfvio_test_sweep_stamps(g.h, 1, None)
for z, R, p in sc.frames(6):
    g.process(sc.dt)
    g.updateWithFeaturePositions(z, R, p)
st = (C.c_int64 * 1024)()
g.lib.ekfvio_test_sweep_stamps(g.h, 1, st)
v = list(st)
names = ["loads+sync", "panel solve", "rank-64 update", "factorisation", "stores"]
for k in range((2 * N + 63) // 64 - 1):
    b = 32 + 8 * k
    print("step %2d: " % k + "  ".join("%s %6d" % (names[i], v[b + i + 1] - v[b + i]) for i in range(5)) + "   | in-kernel total %6d cycles = %.2f us at 2.4 GHz" % (v[b + 5] - v[b], (v[b + 5] - v[b]) / 2400.0))

for k in range((2 * N + 63) // 64 - 1):
    b = 32 + 8 * k
    if v[b + 6]:
        print("step %2d rank-64 update: L_ik store + target requests %5d, product issued over %5d, target tile -> LDS + barrier %5d" % (k, v[b + 7] - v[b + 2], v[b + 6] - v[b + 7], v[b + 3] - v[b + 6]))
for k in range((2 * N + 63) // 64 - 1):
    b = 960 + 4 * k
    print("step %2d panel solve (wave 0): operand preload %5d  MFMA chain %5d  write-back %5d" % (k, v[b + 1] - v[b], v[b + 2] - v[b + 1], v[b + 3] - v[b + 2]))

if v[1000]:
    print("gather+potrf launch, chain workgroup: tile gather %d  factorisation %d  stores %d cycles; last gather workgroup ended %+d, first %+d cycles after the chain workgroup started"
          % (v[1001] - v[1000], v[1002] - v[1001], v[1003] - v[1002], v[1004] - v[1000], v[1005] - v[1000]))

p = v[800:880]
if p[1]:
    print("  first tile's factorisation by panel (cycles): factor phases %s  trails %s  total %d"
          % ([p[2 + 2 * q] - p[1 + 2 * q] for q in range(4)], [p[3 + 2 * q] - p[2 + 2 * q] for q in range(3)], p[10] - p[1]))
