"""N=1024 from the raw prior: cond(S) of the first update is ~3e7, beyond fp32.  What do the HIP path and the fp32
oracle (the reference's arithmetic: LDL^T, which tolerates negative pivots) each do there, and does the filter recover?
Usage (GPU box): python scripts/n1024_prior_check.py [N=1024] [hip_steps=12] [oracle_steps=2]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF, capi  # noqa: E402
from ekf_vio_amd.sim import Scenario  # noqa: E402
from oracle import OracleFilter, max_threads, set_threads  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
hs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
os_ = int(sys.argv[3]) if len(sys.argv) > 3 else 2
set_threads(min(max_threads(), 16))
sc = Scenario(N, seed=0)
g = TightlyCoupledEKF(max_features=N)
g.addNewFeatures(sc.initial_features())
fr = list(sc.frames(hs))
truth = Scenario(N, seed=0)
for s, (z, R, p) in enumerate(fr):
    truth.advance()
    g.process(sc.dt)
    if s == 0:
        S = g.Sigma.astype(np.float64)
        idx = np.array([[22 + 3 * i, 23 + 3 * i] for i in range(N)]).ravel()
        Sm = S[np.ix_(idx, idx)] + 1e-5 * np.eye(2 * N)
        w = np.linalg.eigvalsh(0.5 * (Sm + Sm.T))
        print("first update: eig(S) min %.3e max %.3e cond %.3e" % (w[0], w[-1], w[-1] / w[0]), flush=True)
    rc = g.updateWithFeaturePositions(z, R, p)
    b = g.base_mu
    md, ma = g.checkSigma()
    print("hip step %2d rc %d pos_err %.3e vel_err %.3e min_diag %.3e max_asym %.3e maxabs %.3e" % (
        s, rc, np.abs(b[:3] - truth.pos).max(), np.abs(b[7:10] - truth.vel).max(), md, ma, float(np.abs(g.Sigma).max())), flush=True)
g.close()
truth = Scenario(N, seed=0)
for dtype in (np.float32, np.float64):
    o = OracleFilter(dtype)
    o.add_new_features(sc.initial_features())
    truth = Scenario(N, seed=0)
    for s, (z, R, p) in enumerate(fr[:os_]):
        truth.advance()
        t0 = time.perf_counter()
        o.process(sc.dt)
        rc = o.update(z, R, p)
        st = o.get_state()
        print("oracle %s step %d rc %d pos_err %.3e vel_err %.3e min_diag %.3e maxabs %.3e (%.1f s)" % (
            np.dtype(dtype).name, s, rc, np.abs(st["base_mu"][:3] - truth.pos).max(), np.abs(st["base_mu"][7:10] - truth.vel).max(),
            float(np.diag(st["Sigma"]).min()), float(np.abs(st["Sigma"]).max()), time.perf_counter() - t0), flush=True)
