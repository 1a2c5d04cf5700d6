"""Debug: closed loop from the raw prior, Schur sweep vs GEMM formulation, per step."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF, capi
from ekf_vio_amd.sim import Scenario

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
gs = {}
for mode in ("1", "0"):
    os.environ["EKFVIO_SCHUR"] = mode
    sc = Scenario(N, seed=0)
    g = TightlyCoupledEKF(max_features=N)
    g.addNewFeatures(sc.initial_features())
    gs[mode] = (g, sc, list(sc.frames(steps)))
truth = Scenario(N, seed=0)
for s in range(steps):
    truth.advance()
    line = "step %2d" % s
    sts = {}
    for mode in ("1", "0"):
        g, sc, fr = gs[mode]
        g.process(sc.dt)
        rc = g.updateWithFeaturePositions(*fr[s])
        st = g.get_state()
        sts[mode] = st
        md, ma = g.checkSigma()
        line += " | schur=%s rc %d pos_err %.2e min_diag %.2e asym %.2e maxabs %.2e" % (
            mode, rc, np.abs(st["base_mu"][:3] - truth.pos).max(), md, ma, np.abs(st["Sigma"]).max())
    d = np.abs(sts["1"]["Sigma"].astype(np.float64) - sts["0"]["Sigma"])
    print(line + " | diff %.2e" % d.max(), flush=True)
