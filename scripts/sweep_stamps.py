"""Phase stamps (s_memtime ticks) of the persistent Cholesky sweep during a filter update at N landmarks."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF
from ekf_vio_amd.sim import Scenario
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sc = Scenario(N, seed=0)
g = TightlyCoupledEKF(max_features=N, hooks=True)
g.addNewFeatures(sc.initial_features())
fr = list(sc.frames(6))
g.lib.ekfvio_test_sweep_stamps(g.h, 1, None)
for z, R, p in fr:
    g.process(sc.dt)
    g.updateWithFeaturePositions(z, R, p)
st = (C.c_int64 * 1024)()
g.lib.ekfvio_test_sweep_stamps(g.h, 1, st)
v = list(st)
mb = (2 * N + 63) // 64
print("chain: first potrf+publish %d ticks" % (v[1] - v[0]))
names = ["wait done[k-1]", "load tile", "trsm", "mma+diag update", "potrf", "store+signal"]
for k in range(mb - 1):
    b = 8 + 8 * k
    print("step %2d: " % k + "  ".join("%s %6d" % (names[i], v[b + i + 1] - v[b + i]) for i in range(6)) + "   | total %6d" % (v[b + 6] - v[b]))
print("chain total %d ticks" % (v[8 + 8 * (mb - 2) + 6] - v[0]))
for k in range(mb):
    b = 256 + 8 * k
    if v[b + 2]:
        print("helper1 step %2d: wait done[k-1] %6d  tile loads + wait ready[k] + L loads %6d  work after that %6d  signal %5d   | starts at %7d" %
              (k, v[b + 1] - v[b], v[b + 2] - v[b + 1], v[b + 3] - v[b + 2], v[b + 4] - v[b + 3], v[b] - v[0]))
    else:
        print("helper1 step %2d: wait done[k-1] %6d  all work %6d" % (k, v[b + 1] - v[b], v[b + 3] - v[b + 1]))

# step 2: when did each helper see ready[2] and when did it finish, relative to the chain's ready[2] publication
t_ready2 = v[1008 + 1]  # 100 MHz ticks (10 ns)
t_ready3 = v[1008 + 2]
print('chain: ready[2] -> ready[3] = %d x10ns' % (t_ready3 - t_ready2))
seen = [(v[768 + h] - t_ready2, v[512 + h] - t_ready2, h) for h in range(240) if v[512 + h]]
fin = sorted(x[1] for x in seen)
print("step 2: %d helpers; finish after ready[2]: min %d  median %d  p90 %d  max %d" % (len(fin), fin[0], fin[len(fin) // 2], fin[int(0.9 * len(fin))], fin[-1]))
sw = sorted(x[0] for x in seen if x[0] > -10**9)
print("         saw ready[2]+L loaded after: min %d median %d max %d" % (sw[0], sw[len(sw) // 2], sw[-1]))
slow = sorted(seen, key=lambda x: -x[1])[:8]
print("         slowest helpers (h, saw, fin):", [(x[2], x[0], x[1]) for x in slow])
