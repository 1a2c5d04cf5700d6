"""Debug: symmetric Joseph flow (default) against EKFVIO_JOSEPH_SYM=1: state difference over a device-resident run, status,
timing.  Usage: python scripts/sym_debug.py [N] [steps]"""
import os, sys, time, subprocess, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run(N, steps):
    from ekf_vio_amd import TightlyCoupledEKF, capi
    from ekf_vio_amd.sim import Scenario
    sc = Scenario(N, seed=0)
    g = TightlyCoupledEKF(max_features=N)
    g.addNewFeatures(sc.initial_features())
    fr = list(sc.frames(steps + 10))
    g.upload_measurements(np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]), np.stack([f[2] for f in fr]))
    g.run_uploaded(0, 0, sc.dt)
    out = []
    t0 = time.perf_counter()
    done = 0
    for chunk in (1, 1, 2, 4, 8, 16, steps):
        c = min(chunk, steps - done)
        if c <= 0:
            break
        t1 = time.perf_counter()
        g.run_uploaded(done, c, sc.dt)
        rc = g.synchronize()
        dt = time.perf_counter() - t1
        done += c
        st = g.get_state()
        md, ma = g.checkSigma()
        out.append(dict(steps=done, rc=int(rc), us_per_step=1e6 * dt / c, min_diag=float(md), max_asym=float(ma),
                        pos=[float(x) for x in st["base_mu"][:3]], sig_norm=float(np.linalg.norm(st["Sigma"].astype(np.float64))),
                        finite=bool(np.isfinite(st["Sigma"]).all())))
    np.save("/tmp/sym_debug_%s_%d.npy" % (os.environ.get("EKFVIO_JOSEPH_SYM", "0"), N), st["Sigma"])
    g.close()
    return out


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    if len(sys.argv) > 3:
        print(json.dumps(run(N, steps)))
        sys.exit(0)
    res = {}
    for full in ("1", "0"):
        env = dict(os.environ, EKFVIO_JOSEPH_SYM=full)
        o = subprocess.run([sys.executable, __file__, str(N), str(steps), "child"], env=env, capture_output=True, text=True)
        print("FULL=%s" % full, o.stderr[-500:] if o.returncode else "")
        res[full] = json.loads(o.stdout.strip().splitlines()[-1])
        for r in res[full]:
            print("  ", r)
    a, b = np.load("/tmp/sym_debug_1_%d.npy" % N), np.load("/tmp/sym_debug_0_%d.npy" % N)
    print("final Sigma rel diff", np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(a.astype(np.float64)), "asym full", np.abs(a - a.T).max(), "asym sym", np.abs(b - b.T).max())
