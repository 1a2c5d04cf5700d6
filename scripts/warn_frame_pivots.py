"""in the free-running N = 256 image loop (FAST 20 / 12 px, bench.py's full_loop.n256 settings) the HIP path raises
a pivot warning on one frame where the fp32 oracle loop raises none.  Which frame, which pivot, and how close to zero is the oracle's
pivot there?  The HIP loop runs free; in front of every frame the oracle node is forced to the HIP state, so that its (bit-exact) process(dt)
and tracker hand back exactly the innovation covariance the HIP update factors.  On the flagged frame S is formed on the host as the gather
forms it (A = S^T, lower triangle read), factored by (a) the HIP blocked signed Cholesky (hooks build, ekfvio_test_cholesky_solve: d_c = +-L_cc^2),
(b) an unblocked fp32 LDL^T in the oracle's order, (c) the same in fp64.   Run on the GPU box: python scripts/warn_frame_pivots.py"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ekf_vio_amd import EKFVIO, TightlyCoupledEKF, capi  # noqa: E402
from ekf_vio_amd.sim import translated_sequence  # noqa: E402
from _oracle_node import OracleNode  # noqa: E402
from test_gpu_loop import K, grey  # noqa: E402


def ldlt(A, dtype):
    """unblocked LDL^T of the lower triangle of A, column by column (the order of the oracle's explicit factorisation)"""
    A = np.array(A, dtype=dtype)
    m = A.shape[0]
    L = np.zeros_like(A)
    D = np.zeros(m, dtype)
    for j in range(m):
        v = A[j, j]
        for k in range(j):
            v = dtype(v - dtype(dtype(L[j, k] * L[j, k]) * D[k]))
        D[j] = v
        if j + 1 < m:
            col = A[j + 1:, j].copy()
            for k in range(j):
                col = (col - (L[j + 1:, k] * dtype(L[j, k] * D[k])).astype(dtype)).astype(dtype)
            L[j + 1:, j] = (col / v).astype(dtype)
        L[j, j] = 1
    return D


def main():
    frames = 46
    seq = translated_sequence(grey(), frames)
    v = EKFVIO(max_features=256, replenish=1, fast_threshold=20, min_new_feature_dist=12)
    node = OracleNode(256, K, fast_threshold=20, min_new_feature_dist=12)
    solver = TightlyCoupledEKF(max_features=4, hooks=True)
    out = []
    for i, img in enumerate(seq):
        stamp = 1.0 + i / 30.0
        if i > 0:
            node.ekf.set_state(v.tc_ekf.get_state())  # teacher forcing: the oracle node sees what the HIP loop sees
        rc = v.addFrame(stamp, img, K)
        ro = node.add_frame(stamp, img)
        if rc != capi.ENUMERIC and ro != 1:
            continue
        pre, z, R, p = node.last["pre_update"], node.last["z"], node.last["R"], node.last["passed"]
        idx = np.array([22 + 3 * q + s for q in np.flatnonzero(p) for s in (0, 1)])
        Sig = pre["Sigma"].astype(np.float32)
        m = len(idx)
        S = Sig[np.ix_(idx, idx)].copy()            # S(c, r) = Sigma(idx[c], idx[r]) + R
        for a, q in enumerate(np.flatnonzero(p)):
            S[2 * a, 2 * a] = np.float32(S[2 * a, 2 * a] + R[q, 0])
            S[2 * a + 1, 2 * a + 1] = np.float32(S[2 * a + 1, 2 * a + 1] + R[q, 3])
            S[2 * a + 1, 2 * a] = np.float32(S[2 * a + 1, 2 * a] + R[q, 1])
            S[2 * a, 2 * a + 1] = np.float32(S[2 * a, 2 * a + 1] + R[q, 2])
        A = S.T.copy()                               # what the kernels factor: the lower triangle of S^T
        Lh, _, info = solver.test_cholesky_solve(A, np.zeros((4, m), np.float32))
        dh = np.sign(np.diag(Lh)) * np.diag(Lh).astype(np.float64) ** 2
        d32 = ldlt(A, np.float32).astype(np.float64)
        d64 = ldlt(A.astype(np.float64), np.float64)
        w = np.linalg.eigvalsh((A.astype(np.float64) + A.astype(np.float64).T) / 2)
        line = ("frame %d: HIP status %s, oracle-node status %s (teacher-forced to the HIP state); m = %d, cond(S) = %.3g, min eig (fp64, symmetrised) %.3g\n"
                % (i, "ENUMERIC" if rc == capi.ENUMERIC else "ok", ro, m, w[-1] / max(abs(w[0]), 1e-300), w[0]))
        order = np.argsort(dh)[:4]
        for c in order:
            line += ("   pivot %3d (landmark %d, row %d): HIP blocked factor d = %+.4e | unblocked fp32 LDL^T d = %+.4e | fp64 d = %+.4e | S_cc = %.4e\n"
                     % (c, int(np.flatnonzero(p)[c // 2]), c % 2, dh[c], d32[c], d64[c], float(A[c, c])))
        line += "   non-positive pivots: HIP %d (hook info %d), unblocked fp32 %d, fp64 %d; smallest fp32 / fp64 pivots %.3e / %.3e\n" % (
            int((dh <= 0).sum()), info, int((d32 <= 0).sum()), int((d64 <= 0).sum()), d32.min(), d64.min())
        out.append(line)
        print(line, flush=True)
    if not out:
        print("no frame of the %d raised a warning in either loop" % frames)
    v.tc_ekf.close()


if __name__ == "__main__":
    main()
