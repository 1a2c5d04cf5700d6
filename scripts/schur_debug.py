"""Debug: one update with the Schur sweep vs the GEMM formulation vs the fp64 oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF, capi
from ekf_vio_amd.sim import Scenario
from oracle import OracleFilter

def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-300))

for N in (int(x) for x in (sys.argv[1:] or ["20", "64", "100"])):
    sc = Scenario(N, seed=4)
    o64 = OracleFilter(np.float64)
    o64.add_new_features(sc.initial_features())
    it = sc.frames(8)
    for _ in range(5):
        z, R, p = next(it)
        o64.process(sc.dt), o64.update(z, R, p)
    st = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in o64.get_state().items()}
    z, R, p = next(it)
    out = {}
    for mode in ("1", "0"):
        os.environ["EKFVIO_SCHUR"] = mode
        g = TightlyCoupledEKF(max_features=N)
        g.set_state(st)
        g.process(sc.dt)
        pre = g.get_state()
        rc = g.updateWithFeaturePositions(z, R, p)
        out[mode] = (rc, g.get_state())
        g.close()
    o64.set_state(pre)
    o64.update(z, R, p)
    s64 = o64.get_state()
    for mode in ("1", "0"):
        rc, s = out[mode]
        d = np.abs(s["Sigma"].astype(np.float64) - s64["Sigma"])
        i, j = np.unravel_index(np.argmax(d), d.shape)
        print("N=%d schur=%s rc=%d Sigma rel %.3e maxabs %.3e at (%d,%d) base_mu maxabs %.3e feat %.3e" % (
            N, mode, rc, rel(s["Sigma"], s64["Sigma"]), d.max(), i, j,
            np.abs(s["base_mu"] - s64["base_mu"]).max(), np.abs(s["feat_mu"] - s64["feat_mu"]).max()))
        if mode == "1":
            blk = d.reshape(d.shape) 
            nb = (d.shape[0] + 63) // 64
            e = np.zeros((nb, nb))
            for a in range(nb):
                for b in range(nb):
                    e[a, b] = d[a*64:(a+1)*64, b*64:(b+1)*64].max()
            print(np.array2string(e, precision=1, max_line_width=200))
