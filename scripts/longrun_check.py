import numpy as np, sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF
from ekf_vio_amd.sim import Scenario
from oracle import OracleFilter, set_threads
set_threads(1)  # the oracle's OpenMP products crawl with a big host's default thread count at this size
N = 64
sc = Scenario(N, seed=3)
fr = list(sc.frames(600))
g = TightlyCoupledEKF(max_features=N)
g.addNewFeatures(sc.initial_features())
o = OracleFilter(np.float32)
o.add_new_features(sc.initial_features())
o64 = OracleFilter(np.float64)
o64.add_new_features(sc.initial_features())
for k, (z, R, p) in enumerate(fr):
    g.process(sc.dt); g.updateWithFeaturePositions(z, R, p)
    o.process(sc.dt); o.update(z, R, p)
    o64.process(sc.dt); o64.update(z, R, p)
    if k in (9, 49, 99, 199, 399, 599):
        for name, f in (("gpu", g.get_state()), ("orc32", o.get_state()), ("orc64", o64.get_state())):
            S = f["Sigma"].astype(np.float64)
            asym = np.abs(S - S.T)
            i, j = np.unravel_index(np.argmax(asym), asym.shape)
            print(k + 1, name, "max|S| %.3e" % np.abs(S).max(), "max diag %.3e" % S.diagonal().max(), "min diag %.3e" % S.diagonal().min(),
                  "max asym %.3e at (%d,%d) rel %.2e" % (asym.max(), i, j, asym.max() / max(abs(S[i, j]), 1e-30)), "pos", f["base_mu"][:3], flush=True)
