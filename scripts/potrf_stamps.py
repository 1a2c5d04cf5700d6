import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF
g = TightlyCoupledEKF(max_features=4, hooks=True)
st = (C.c_int64 * 80)()
import sys as _s
reps = int(_s.argv[1]) if len(_s.argv) > 1 else 3
for _rep in range(reps):   # the first launch runs with a cold instruction cache
    print("rc", g.lib.ekfvio_test_potrf_stamps(g.h, st))
    v = list(st)
    print("launch %d total %d ticks" % (_rep, v[11] - v[0]))
names = ["load", "p0 factor", "p0 trail", "p1 factor", "p1 trail", "p2 factor", "p2 trail", "p3 factor", "p3 trail", "zero+inverse", "store"]
# stamps: 0 start,1 after load(+sync) ,2.. 
for i in range(1, 12):
    print("%-14s %7d ticks" % (names[i - 1], v[i] - v[i - 1]))
print("panel 0 chain (between the loads and the stores): %d ticks; F0 opened at %d, chain began +%d" % (v[15]-v[14], v[1], v[14]-v[1]))
print("hash L %016x  inv %016x" % (v[12] & (2**64-1), v[13] & (2**64-1)))
print("total", v[11] - v[0], "ticks (s_memtime = 100 MHz constant clock? see below)")

ph = ["F0", "C0", "F1", "C1", "F2", "C2", "F3", "-", "fin"]
print("per-wave end-of-work (ticks after the phase's opening barrier stamp):")
opening = [v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9]]
for i, nm in enumerate(ph):
    if nm == "-":
        continue
    print("  %-4s" % nm, " ".join("w%d %6d" % (w, v[16 + 16 * w + i] - opening[i]) if v[16 + 16 * w + i] else "w%d      -" % w for w in range(4)))
