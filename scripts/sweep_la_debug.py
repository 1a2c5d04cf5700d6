"""Split sweep, look-ahead form against the two-launch form and numpy: factor L and solve X = C S^-1 at m = 1024 (16 block steps)."""
import os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 1:
    from ekf_vio_amd import TightlyCoupledEKF
    m, nr = int(sys.argv[2]), int(sys.argv[3])
    rng = np.random.default_rng(5)
    A = rng.standard_normal((m, m)).astype(np.float32)
    S = (A @ A.T / m + np.eye(m, dtype=np.float32)).astype(np.float32)
    Cr = rng.standard_normal((nr, m)).astype(np.float32)
    g = TightlyCoupledEKF(max_features=m // 2, hooks=True)
    L, X, info = g.test_cholesky_solve(S, Cr)
    np.save(sys.argv[1], np.concatenate([L.ravel(), X.ravel(), [info]]))
    Lr = np.linalg.cholesky(S.astype(np.float64))
    Xr = Cr.astype(np.float64) @ np.linalg.inv(S.astype(np.float64))
    print("info", info, "max |L - L64|", float(np.abs(np.tril(L) - Lr).max()), "max |X - X64|", float(np.abs(X - Xr).max()))
    d = np.abs(np.tril(L) - Lr).reshape(m // 64, 64, m // 64, 64).max(axis=(1, 3))
    print("block errors of L (rows = row block):")
    for row in d[:8, :8]:
        print(" ".join("%8.1e" % v for v in row))
    sys.exit(0)
for m, nr in ((1024, 200), (1088, 70)):
    outs = []
    for la in ("0", "1"):
        env = dict(os.environ, EKFVIO_SWEEP_LA=la)
        out = "/tmp/la_%s.npy" % la
        print("LA", la, subprocess.run([sys.executable, __file__, out, str(m), str(nr)], env=env, capture_output=True, text=True).stdout)
        outs.append(np.load(out))
    print("m %d: identical bits: %s" % (m, np.array_equal(outs[0], outs[1])))
