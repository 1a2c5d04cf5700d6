import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF, capi
from ekf_vio_amd.sim import Scenario
N = 256
sc = Scenario(N, seed=0)
g = TightlyCoupledEKF(max_features=N)
g.addNewFeatures(sc.initial_features())
import sys as _s
ST = int(_s.argv[1]) if len(_s.argv) > 1 else 320
fr = list(sc.frames(ST + 120))
g.upload_measurements(np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]), np.stack([f[2] for f in fr]))
def percall(tag, a, b):
    t0 = time.perf_counter(); rcs = []
    for z, R, p in fr[a:b]:
        g.process(sc.dt); rcs.append(g.updateWithFeaturePositions(z, R, p))
    print(tag, "%.0f steps/s" % ((b - a) / (time.perf_counter() - t0)), "warnings", sum(r != 0 for r in rcs), g.sweep_counts(), flush=True)
g.run_uploaded(0, 0, sc.dt); g.run_uploaded(0, ST, sc.dt); print("sync", g.synchronize())
percall("after graph run", ST, ST + 20)
g.profile(True); g.run_uploaded(ST + 20, 20, sc.dt); g.synchronize(); g.profile(False)
percall("after profile", ST + 40, ST + 60)
print(g.profile_update_gemms(50))
percall("after profile_update_gemms", ST + 60, ST + 80)
