"""Round 6: where the fused persistent launch spends its time with T2 = Sigma (I - K H)^T formed inside (EKFVIO_T2=1) and without (=0): the chain's
block steps, the gain tiles' ends, three T2 tile pairs' block columns, the last workgroup's exit -- all on the 100 MHz s_memrealtime clock (one clock
for every XCD), in us from the chain workgroup's first stamp.  The stamps are those of the LAST step of a graph replay (steady state), argv[2] = 0: of
a per-call update."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF
from ekf_vio_amd.sim import Scenario
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
replay = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sc = Scenario(N, seed=0)
g = TightlyCoupledEKF(max_features=N, hooks=True)
g.addNewFeatures(sc.initial_features())
g.lib.ekfvio_test_sweep_stamps(g.h, 1, None)
fr = list(sc.frames(40))
if replay:
    g.upload_measurements(np.stack([f[0] for f in fr]).astype(np.float32), np.stack([f[1] for f in fr]).astype(np.float32), np.stack([f[2] for f in fr]).astype(np.uint8))
    g.run_uploaded(0, 0, sc.dt)
    g.run_uploaded(0, 34, sc.dt)
    g.synchronize()
else:
    for z, R, p in fr[:6]:
        g.process(sc.dt)
        g.updateWithFeaturePositions(z, R, p)
st = (C.c_int64 * 1024)()
g.lib.ekfvio_test_sweep_stamps(g.h, 1, st)
v = list(st)
mb = (2 * N + 63) // 64
t0 = v[896]
us = lambda x: (x - t0) / 100.0 if x else float("nan")
print("EKFVIO_T2 =", os.environ.get("EKFVIO_T2", "default"), " N =", N, "graph replay" if replay else "per call")
print("chain: factorisation of step k ends at", ["%.1f" % us(v[700 + k]) for k in range(mb - 1)], " chain ends at %.1f us" % us(v[897]))
nX = (22 + 3 * N + 63) // 64
print("gain tiles end at (row block x block column), and their last block column's operands were solved at:")
for ib in range(nX):
    print("   ", " ".join("%5.1f" % us(v[200 + ib * mb + cb]) for cb in range(mb)), "  |  ", " ".join("%5.1f" % us(v[400 + ib * mb + cb]) for cb in range(mb)))
for w, name in enumerate(("(0,0)", "(0,3)", "(0,7)", "(6,0)", "(6,3)", "(6,7)")):
    if v[100 + 8 * w + mb - 1]:
        print("gain tile %s: block columns done at %s" % (name, ["%.1f" % us(v[100 + 8 * w + k]) for k in range(mb)]))
if v[304]:
    print("T2 pairs end at:", " ".join("%.1f" % us(v[304 + p]) for p in range(nX * (nX + 1) // 2)))
for w, name in enumerate(("first", "middle", "last")):
    b = 840 + 16 * w
    if v[b + 14]:
        print("T2 pair %-6s adopted at %.1f, block columns done at %s, stored at %.1f" % (name, us(v[b + 14]), ["%.1f" % us(v[b + k]) for k in range(mb)], us(v[b + 15])))
print("blocks 252..267 start at", ["%.1f/xcc%d" % (us(v[160 + i]), v[176 + i]) for i in range(16) if v[160 + i]])
print("owners 0..7 are done with their tile at", ["%.1f" % us(v[192 + i]) for i in range(8) if v[192 + i]])
print("the launch's last workgroup (block %d) leaves at %.1f us" % (v[899], us(v[898])))
g.close()
