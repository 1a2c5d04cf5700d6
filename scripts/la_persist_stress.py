"""Stress of the split sweep's persistent form (EKFVIO_SWEEP_LA_PERSIST=1, chol_persist_la.inc) against one launch per block step: many fresh
handles, several sizes, other work in between (the rare cache-coherence faults of the first versions showed only inside full test runs)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from ekf_vio_amd import TightlyCoupledEKF
from ekf_vio_amd.sim import Scenario
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for rep in range(reps):
    for N, fails in ((700, 19), (544, 0), (1024, 5)):
        sc = Scenario(N, seed=13 + rep)
        fr = list(sc.frames(3))
        for s, (z, R, p) in enumerate(fr):
            for q in range(fails):
                p[(7 * q + 3 * s + 1 + rep) % N] = 0
        out = {}
        for mode in ("0", "1"):
            os.environ["EKFVIO_SWEEP_LA_PERSIST"] = mode
            g = TightlyCoupledEKF(max_features=N)
            g.addNewFeatures(sc.initial_features())
            for z, R, p in fr:
                g.process(sc.dt)
                g.updateWithFeaturePositions(z, R, p)
            out[mode] = g.get_state()
            assert (g.persistent_sweeps() > 0) == (mode == "1")
            g.close()
        same = all(np.array_equal(out["0"][k], out["1"][k]) for k in ("base_mu", "feat_mu", "Sigma"))
        bad += not same
        if not same:
            print("rep", rep, "N", N, "DIFFERS", flush=True)
    # other work in between: a small filter with the N = 256 persistent sweep
    os.environ.pop("EKFVIO_SWEEP_LA_PERSIST", None)
    sc = Scenario(256, seed=rep)
    g = TightlyCoupledEKF(max_features=256)
    g.addNewFeatures(sc.initial_features())
    for z, R, p in sc.frames(3):
        g.process(sc.dt)
        g.updateWithFeaturePositions(z, R, p)
    g.close()
print("%d repetitions x 3 sizes: %d differences" % (reps, bad))
