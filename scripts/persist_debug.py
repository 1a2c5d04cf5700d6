"""Which tiles of the factor / of the solve differ between the per-step sweep and the persistent per-tile sweep."""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from ekf_vio_amd import TightlyCoupledEKF
    m, nr = int(sys.argv[2]), int(sys.argv[3])
    rng = np.random.default_rng(7)
    Q = rng.standard_normal((m, m))
    S = (Q @ Q.T / m + np.eye(m) * 0.1).astype(np.float32)
    Cr = rng.standard_normal((nr, m)).astype(np.float32)
    g = TightlyCoupledEKF(max_features=4, hooks=True)
    L, X, info = g.test_cholesky_solve(S, Cr)
    np.save(sys.argv[4] + "_L.npy", L); np.save(sys.argv[4] + "_X.npy", X)
    print("info", info)
    sys.exit(0)
m, nr = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 150)
for mode in ("0", "2"):
    env = dict(os.environ, EKFVIO_SWEEP=mode, EKFVIO_DEBUG_DUMP_LAUG="/tmp/pd_%s_aug.bin" % mode)
    out = subprocess.run([sys.executable, __file__, "child", str(m), str(nr), "/tmp/pd_" + mode], env=env, capture_output=True, text=True)
    print("mode", mode, out.stdout.strip(), out.stderr.strip()[-300:])
F0, F2 = np.load("/tmp/pd_0_L.npy"), np.load("/tmp/pd_2_L.npy")
L0, L2 = np.tril(F0), np.tril(F2)
X0, X2 = np.load("/tmp/pd_0_X.npy"), np.load("/tmp/pd_2_X.npy")
mb = (m + 63) // 64
print("L tiles that differ (row, col): max abs diff")
for i in range(mb):
    print(" ".join("%9.2e" % np.abs(L0[64 * i:64 * i + 64, 64 * j:64 * j + 64] - L2[64 * i:64 * i + 64, 64 * j:64 * j + 64]).max() if j <= i else "    -    " for j in range(mb)))
print("X = C S^-1 column blocks: max abs diff", [float(np.abs(X0[:, 64 * j:64 * j + 64] - X2[:, 64 * j:64 * j + 64]).max()) for j in range(mb)])
print("X row blocks:", [float(np.abs(X0[64 * i:64 * i + 64] - X2[64 * i:64 * i + 64]).max()) for i in range((nr + 63) // 64)])


def aug(path):
    h = np.fromfile(path, dtype=np.int32, count=3)
    ld, mp, rp = int(h[0]), int(h[1]), int(h[2])
    a = np.fromfile(path, dtype=np.float32, offset=12).reshape(mp, ld).T  # column-major ld x mp
    return a, ld, mp, rp
A0, ld, mp, rp = aug("/tmp/pd_0_aug.bin")
A2 = aug("/tmp/pd_2_aug.bin")[0]
print("whole swept matrix, row blocks x block columns (A rows, then X, then identity): max abs diff; identity blocks right of their diagonal are unused")
for i in range(ld // 64):
    kind = "A" if i < mp // 64 else ("X" if i < (mp + rp) // 64 else "I")
    print(" %s%2d " % (kind, i) + " ".join("%9.2e" % np.abs(A0[64 * i:64 * i + 64, 64 * j:64 * j + 64] - A2[64 * i:64 * i + 64, 64 * j:64 * j + 64]).max() for j in range(mp // 64)))
