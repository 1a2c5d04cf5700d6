"""The split sweep as one persistent launch (chol_persist_la.inc) against one launch per block step (EKFVIO_SWEEP=0): bits and time.
usage: python scripts/la_persist_debug.py [N] [steps] [fails]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
fails = int(sys.argv[3]) if len(sys.argv) > 3 else 0
from ekf_vio_amd import TightlyCoupledEKF, capi
from ekf_vio_amd.sim import Scenario
sc = Scenario(N, seed=5)
fr = list(sc.frames(steps))
for s, (z, R, p) in enumerate(fr):
    for q in range(fails):
        p[(7 * q + 3 * s + 1) % N] = 0
out = {}
for mode in ("0", "2"):
    os.environ["EKFVIO_SWEEP"] = mode
    g = TightlyCoupledEKF(max_features=N)
    g.addNewFeatures(sc.initial_features())
    rcs = []
    for z, R, p in fr:
        g.process(sc.dt)
        rcs.append(g.updateWithFeaturePositions(z, R, p))
    out[mode] = g.get_state()
    print("mode", mode, "rc", rcs, "sweep counts", g.sweep_counts(), flush=True)
    # device-resident timing
    g.upload_measurements(np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]), np.stack([f[2] for f in fr]))
    g.run_uploaded(0, len(fr), sc.dt); g.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        g.run_uploaded(0, len(fr), sc.dt)
    g.synchronize()
    dt = (time.perf_counter() - t0) / (reps * len(fr))
    print("mode", mode, "%.1f us per step (replayed, state not restored)" % (dt * 1e6), "counts", g.sweep_counts(), flush=True)
    g.close()
ok = True
for k in ("base_mu", "feat_mu", "Sigma", "last_klt", "del_flag"):
    same = np.array_equal(out["0"][k], out["2"][k])
    ok &= same
    if not same:
        d = np.abs(out["0"][k].astype(np.float64) - out["2"][k].astype(np.float64))
        print(k, "DIFFERS max", np.nanmax(d), "count", int((d > 0).sum()), "nan", int(np.isnan(d).sum()))
print("bit-identical" if ok else "MISMATCH")
