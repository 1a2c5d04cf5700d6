"""In-kernel phase stamps (s_memtime cycles) of predict_fused_kernel<true> during a filter step at N landmarks."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF
from ekf_vio_amd.sim import Scenario
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sc = Scenario(N, seed=0)
g = TightlyCoupledEKF(max_features=N, hooks=True)
g.addNewFeatures(sc.initial_features())
g.lib.ekfvio_test_sweep_stamps(g.h, 1, None)
for z, R, p in sc.frames(5):
    g.process(sc.dt)
    g.updateWithFeaturePositions(z, R, p)
st = (C.c_int64 * 1024)()
g.lib.ekfvio_test_sweep_stamps(g.h, 1, st)
v = list(st)
names = ["diagonal tile (0,0)", "tile (0,1)", "base workgroup 0 (22x22)", "base rows, chunk 0", "base columns, chunk 0"]
for w in range(5):
    b = 900 + 8 * w
    s = [v[b + i] for i in range(5)]
    if w < 2:
        print("%-26s loads + base motions %5d  derivative columns %5d  X on base columns %5d  3x3 blocks (P loads, products, stores) %5d  | total %5d cycles" % (
            names[w], s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[4] - s[0]))
    else:
        print("%-26s Jacobian A (+ blocks) %5d  products and stores %5d  | total %5d cycles" % (names[w], s[1] - s[0], s[4] - s[1], s[4] - s[0]))
