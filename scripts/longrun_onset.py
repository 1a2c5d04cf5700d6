"""Where does the fp32 filter give way on the long synthetic run?  max |Sigma| of the HIP filter, the fp32 oracle and the fp64
oracle every 20 steps (N and seed from the command line)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF
from ekf_vio_amd.sim import Scenario
from oracle import OracleFilter, set_threads
set_threads(1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 700
sc = Scenario(N, seed=seed)
g = TightlyCoupledEKF(max_features=N)
g.addNewFeatures(sc.initial_features())
o = OracleFilter(np.float32); o.add_new_features(sc.initial_features())
o64 = OracleFilter(np.float64); o64.add_new_features(sc.initial_features())
for k, (z, R, p) in enumerate(sc.frames(steps)):
    rc = g.process(sc.dt); rc = g.updateWithFeaturePositions(z, R, p)
    o.process(sc.dt); o.update(z, R, p)
    o64.process(sc.dt); o64.update(z, R, p)
    if (k + 1) % 20 == 0 and k + 1 >= 300:
        a, b, c = g.get_state()["Sigma"], o.get_state()["Sigma"], o64.get_state()["Sigma"]
        print("%4d  max|S| hip %.3e  orc32 %.3e  orc64 %.3e   min diag hip %.2e orc32 %.2e   update rc %d" %
              (k + 1, np.abs(a).max(), np.abs(b).max(), np.abs(c).max(), a.diagonal().min(), b.diagonal().min(), rc), flush=True)
g.close()
