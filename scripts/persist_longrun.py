"""Long graph-replayed runs with the persistent sweep against the per-step sweep: the final state must agree bit for bit, run
after run (a timing-dependent hand-off bug would show as a difference in some repetition).  usage: persist_longrun.py [N] [steps] [reps]"""
import hashlib, os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from ekf_vio_amd import TightlyCoupledEKF
    from ekf_vio_amd.sim import Scenario
    N, steps = int(sys.argv[2]), int(sys.argv[3])
    sc = Scenario(N, seed=5)
    fr = list(sc.frames(64))
    g = TightlyCoupledEKF(max_features=N)
    g.addNewFeatures(sc.initial_features())
    g.upload_measurements(np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]), np.stack([f[2] for f in fr]))
    g.run_uploaded(0, steps, sc.dt)
    g.synchronize()
    st = g.get_state()
    h = hashlib.sha256()
    for k in ("base_mu", "feat_mu", "Sigma"):
        h.update(np.ascontiguousarray(st[k]).tobytes())
    print(h.hexdigest()[:16], "finite" if np.isfinite(st["Sigma"]).all() else "NOT FINITE", "persistent sweeps", g.persistent_sweeps())
    sys.exit(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ref = None
for mode in ["0"] + ["1"] * reps:
    env = dict(os.environ, EKFVIO_SWEEP=mode)
    out = subprocess.run([sys.executable, __file__, "child", str(N), str(steps)], env=env, capture_output=True, text=True)
    line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr.strip()[-300:]
    print("EKFVIO_SWEEP=%s N=%d steps=%d: %s" % (mode, N, steps, line), flush=True)
    digest = line.split()[0]
    if ref is None:
        ref = digest
    elif digest != ref:
        print("DIFFERENT from the per-step sweep")
        sys.exit(1)
print("all runs agree")
