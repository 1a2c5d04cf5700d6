"""Round 6: how far from positive semi-definite does each flow of the update leave the covariance?  Free run from the raw prior (cond(S) ~ 1e6 .. 1e7 in
the first update), min eigenvalue of sym(Sigma) and checkSigma's asymmetry after each of the first steps, two-GEMM flow (the reference's
operation order: a congruence with ONE K, positive semi-definite whatever K is) against the T2 flow.  usage: python scripts/t2_psd_check.py N [steps] [R]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF
from ekf_vio_amd.sim import Scenario
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mv = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-5
for name, env in (("two-GEMM", {"EKFVIO_T2": "0", "EKFVIO_T2_SYRK": "0"}), ("T2 flow", {"EKFVIO_T2": "1", "EKFVIO_T2_SYRK": "1"})):
    os.environ.update(env)
    sc = Scenario(N, seed=0, meas_var=mv)
    g = TightlyCoupledEKF(max_features=N)
    g.addNewFeatures(sc.initial_features())
    for s, (z, R, p) in enumerate(sc.frames(steps)):
        g.process(sc.dt)
        rc = g.updateWithFeaturePositions(z, R, p)
        S = g.Sigma.astype(np.float64)
        w = np.linalg.eigvalsh(0.5 * (S + S.T))
        print("N=%d R=%g %-8s step %d rc %d  eig(sym Sigma) min %.3e (#neg %d)  max asym %.3e" % (N, mv, name, s, rc, w[0], int((w < 0).sum()), g.checkSigma()[1]), flush=True)
    g.close()
