"""In-kernel phase stamps (s_memtime cycles) of the chain workgroup of chol_persist_kernel (EKFVIO_SWEEP=2) during a filter update."""
import ctypes as C, os, sys
os.environ.setdefault("EKFVIO_SWEEP", "2")
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
names = ["operands, look, tile requests, publish", "pin", "panel solve", "update, block column 0", "factorisation", "stores"]
mb = (2 * N + 63) // 64
t0 = v[0]
if v[1] and v[3]:  # fused launch (round 4): the chain workgroup's own prologue
    print("fused launch, chain workgroup: tile (0,0) gather %d  factorisation %d  stores + ready[0] + wait for the gather workgroups %d  (step 0 starts %d cycles after the launch's first stamp)"
          % (v[1] - v[0], v[2] - v[1], v[3] - v[2], v[32] - v[0]))
    print("tile (0,0): the one-batch guess (all of the first 32 landmarks measured) %s" % ("held" if v[900] else "did NOT hold: gathered again"))
for k in range(mb - 1):
    b = 32 + 8 * k
    e = [v[b + i] for i in range(6)] + [v[b + 8]]
    print("step %2d at %6d: " % (k, e[0] - t0) + "  ".join("%s %5d" % (names[i], e[i + 1] - e[i]) for i in range(6))
          + "   | %6d cycles = %.2f us at 2.4 GHz" % (e[6] - e[0], (e[6] - e[0]) / 2400.0))
end = v[32 + 8 * (mb - 1)]
print("chain total %d cycles = %.2f us" % (end - t0, (end - t0) / 2400.0))
hn = ["waits for ready / fin", "loads (one round trip) + barrier", "substitutions", "product", "tile -> LDS, store, publish"]
for j in range(2, mb - 1):
    b = 600 + 8 * j
    if v[b]:  # (these run on other XCDs: their s_memtime counters have other origins, only differences mean something)
        print("helper of tile (%d,%d), its last step (what the chain's step %d waits for): " % (j + 1, j, j)
              + "  ".join("%s %5d" % (hn[q], v[b + q + 1] - v[b + q]) for q in range(5)))
sn = ["wait sources", "load sources", "wait readyA", "L_kk + stage A", "partial product", "wait ready", "stage B + tail"]
for j in range(2, mb - 1):
    b = 760 + 8 * (j & 7)
    if v[b] and v[b + 7]:
        print("staged last step of tile (%d,%d): " % (j + 1, j) + "  ".join("%s %5d" % (sn[q], v[b + q + 1] - v[b + q]) for q in range(7)))
print("wavefront 0's early look at the step's two flags (100 = both up, 101 = not yet):", [int(v[500 + k]) for k in range(1, mb - 1)])
print("slack of the chain's look at step k's end (for tiles (k+2,k+1) / (k+2,k+2)) behind their owners' publish, cycles at 2.4 GHz (100 MHz clock: +-24):")
print([(int((v[700 + k] - v[720 + k + 1]) * 24) if v[720 + k + 1] else None, int((v[700 + k] - v[740 + k + 2]) * 24) if v[740 + k + 2] else None) for k in range(mb - 2)])
