"""Round 6: one update whose innovation covariance is numerically indefinite (the factorisation meets negative pivots and goes through the signed
factor), under the two-GEMM flow and the T2 flow, against the fp32 and fp64 oracles from the same fp32 state.  The state: N landmarks, R = 1e-8,
in front of the first flagged update of the two-GEMM flow's free run from the raw prior.  usage: python scripts/t2_indefinite_check.py [N=256] [R=1e-8]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF, capi
from ekf_vio_amd.sim import Scenario
from oracle import OracleFilter, set_threads, max_threads
set_threads(min(max_threads(), 16))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mv = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8
sc = Scenario(N, seed=0, meas_var=mv)
frames = list(sc.frames(8))
os.environ["EKFVIO_T2"] = "0"
os.environ["EKFVIO_T2_SYRK"] = "0"
g = TightlyCoupledEKF(max_features=N)
g.addNewFeatures(sc.initial_features())
st, zRp = None, None
for s, (z, R, p) in enumerate(frames):
    g.process(sc.dt)
    before = g.get_state()
    rc = g.updateWithFeaturePositions(z, R, p)
    if rc == capi.ENUMERIC:
        st, zRp = before, (z, R, p)
        print("first flagged update: step", s)
        break
g.close()
if st is None:
    print("no flagged update in", len(frames), "steps")
    sys.exit(0)
z, R, p = zRp
o32, o64 = OracleFilter(np.float32), OracleFilter(np.float64)
o32.set_state(st), o64.set_state(st)
i32, i64 = o32.update(z, R, p), o64.update(z, R, p)
s32, s64 = o32.get_state(), o64.get_state()
ma = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())
rf = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(np.asarray(b, np.float64)))
print("oracle fp32 status %d (fp64 %d): |o32 - f64| base %.3e landmarks %.3e Sigma rel %.3e" % (i32, i64, ma(s32["base_mu"], s64["base_mu"]), ma(s32["feat_mu"], s64["feat_mu"]), rf(s32["Sigma"], s64["Sigma"])))
for name, env in (("two-GEMM flow", {"EKFVIO_T2": "0", "EKFVIO_T2_SYRK": "0"}), ("T2 flow", {"EKFVIO_T2": "1", "EKFVIO_T2_SYRK": "1"})):
    os.environ.update(env)
    g = TightlyCoupledEKF(max_features=N)
    g.set_state(st)
    rc = g.updateWithFeaturePositions(z, R, p)
    sg = g.get_state()
    md, masym = g.checkSigma()
    print("%-14s rc %d t2_updates %d: |hip - f64| base %.3e landmarks %.3e Sigma rel %.3e   min diag %.3e max asym %.3e" % (
        name, rc, g.counters()["t2_updates"], ma(sg["base_mu"], s64["base_mu"]), ma(sg["feat_mu"], s64["feat_mu"]), rf(sg["Sigma"], s64["Sigma"]), md, masym))
    g.close()
