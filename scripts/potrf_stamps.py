import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF
g = TightlyCoupledEKF(max_features=4)
st = (C.c_int64 * 12)()
print("rc", g.lib.ekfvio_test_potrf_stamps(g.h, st))
v = list(st)
names = ["load", "p0 factor", "p0 trail", "p1 factor", "p1 trail", "p2 factor", "p2 trail", "p3 factor", "p3 trail", "zero+inverse", "store"]
# stamps: 0 start,1 after load(+sync) ,2.. 
for i in range(1, 12):
    print("%-14s %7d ticks" % (names[i - 1], v[i] - v[i - 1]))
print("total", v[11] - v[0], "ticks (s_memtime = 100 MHz constant clock? see below)")
