"""GPU diagnostic: measures HIP-vs-oracle gaps next to the oracle's own fp32-vs-fp64 gaps
(teacher-forced single steps and free-running loops).  Output feeds the tolerances written
into tests/test_gpu_parity.py.  Run on the GPU box: python scripts/gpu_diag.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF  # noqa: E402
from ekf_vio_amd.sim import Scenario  # noqa: E402
from oracle import OracleFilter  # noqa: E402


def relf(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) /
                 max(np.linalg.norm(np.asarray(b, np.float64)), 1e-300))


def maxabs(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max()) if np.size(a) else 0.0


def gaps(tag, g, o32, o64):
    sg, s32, s64 = g.get_state(), o32.get_state(), o64.get_state()
    return {"tag": tag,
            "mu_gpu_vs_o32": maxabs(sg["base_mu"], s32["base_mu"]), "mu_o32_vs_o64": maxabs(s32["base_mu"], s64["base_mu"]),
            "feat_gpu_vs_o32": maxabs(sg["feat_mu"], s32["feat_mu"]), "feat_o32_vs_o64": maxabs(s32["feat_mu"], s64["feat_mu"]),
            "sig_relF_gpu_vs_o32": relf(sg["Sigma"], s32["Sigma"]), "sig_relF_o32_vs_o64": relf(s32["Sigma"], s64["Sigma"]),
            "sig_relF_gpu_vs_o64": relf(sg["Sigma"], s64["Sigma"]),
            "sig_maxabs_gpu_vs_o32": maxabs(sg["Sigma"], s32["Sigma"]), "sig_maxabs_o32_vs_o64": maxabs(s32["Sigma"], s64["Sigma"]),
            "bookkeeping_equal": bool(np.array_equal(sg["del_flag"], s32["del_flag"]) and np.array_equal(sg["last_klt"], s32["last_klt"]))}


def main():
    out = []
    # raw GEMM
    g = TightlyCoupledEKF(max_features=64, hooks=True)
    rng = np.random.default_rng(0)
    for (M, N, K, tb) in [(64, 64, 16, True), (130, 70, 48, True), (200, 64, 64, False), (790, 790, 512, True), (257, 129, 80, False)]:
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
        C0 = rng.standard_normal((M, N)).astype(np.float32)
        Cg = g.test_gemm(A, B, C0, alpha=-1.0, beta=1.0, transB=tb)
        ref = C0.astype(np.float64) - A.astype(np.float64) @ (B.T if tb else B).astype(np.float64)
        out.append({"tag": "gemm %dx%dx%d tb=%d" % (M, N, K, tb), "maxabs": maxabs(Cg, ref), "rel": relf(Cg, ref)})
        print(out[-1], flush=True)
    for m, nr in [(64, 30), (128, 100), (200, 214), (512, 790)]:
        Q = rng.standard_normal((m, m))
        S = (Q @ Q.T / m + np.eye(m) * 0.1).astype(np.float32)
        Cr = rng.standard_normal((nr, m)).astype(np.float32)
        L, X, info = g.test_cholesky_solve(S, Cr)
        Lref = np.linalg.cholesky(S.astype(np.float64))
        Xref = Cr.astype(np.float64) @ np.linalg.inv(S.astype(np.float64))
        out.append({"tag": "chol m=%d nrhs=%d" % (m, nr), "L_rel": relf(np.tril(L), Lref), "X_rel": relf(X, Xref), "info": info})
        print(out[-1], flush=True)
    g.close()

    for N, steps in [(3, 5), (30, 20), (100, 30)]:
        sc = Scenario(N, seed=0, dt=0.05)
        g = TightlyCoupledEKF(max_features=N, hooks=True)
        o32, o64 = OracleFilter(np.float32), OracleFilter(np.float64)
        uv = sc.initial_features()
        g.addNewFeatures(uv), o32.add_new_features(uv), o64.add_new_features(uv)
        out.append(gaps("init N=%d" % N, g, o32, o64))
        Fg, Fo = g.numericallyLinearizeProcess(sc.dt), o32.linearize(sc.dt)
        out.append({"tag": "linearize init N=%d" % N, "F_maxabs": maxabs(Fg, Fo), "F_equal": bool(np.array_equal(Fg, Fo))})
        print(out[-1], flush=True)
        tf = []
        for s, (z, R, p) in enumerate(sc.frames(steps)):
            if N >= 30 and s % 3 == 1:
                p = p.copy()
                p[(s * 7) % N] = 0
            # teacher-forced: load the fp32 oracle state into the GPU and the fp64 oracle
            st = o32.get_state()
            g.set_state(st)
            o64.set_state(st)
            Fg, Fo = g.numericallyLinearizeProcess(sc.dt), o32.linearize(sc.dt)
            feq = bool(np.array_equal(Fg, Fo))
            fmax = maxabs(Fg, Fo)
            g.process(sc.dt), o32.process(sc.dt), o64.process(sc.dt)
            a = gaps("tf process N=%d s=%d" % (N, s), g, o32, o64)
            a["F_equal"], a["F_maxabs"] = feq, fmax
            st = o32.get_state()
            g.set_state(st)
            o64.set_state(st)
            rc = g.updateWithFeaturePositions(z, R, p)
            o32.update(z, R, p), o64.update(z, R, p)
            b = gaps("tf update N=%d s=%d" % (N, s), g, o32, o64)
            b["rc"] = rc
            tf += [a, b]
        out += tf
        for r in tf[:4] + tf[-4:]:
            print(r, flush=True)
        # free running from scratch
        sc = Scenario(N, seed=1, dt=0.05)
        g.initializeBaseState()
        o32, o64 = OracleFilter(np.float32), OracleFilter(np.float64)
        uv = sc.initial_features()
        g.addNewFeatures(uv), o32.add_new_features(uv), o64.add_new_features(uv)
        for s, (z, R, p) in enumerate(sc.frames(steps)):
            g.process(sc.dt), o32.process(sc.dt), o64.process(sc.dt)
            g.updateWithFeaturePositions(z, R, p), o32.update(z, R, p), o64.update(z, R, p)
            if s in (0, 1, 4, 9, 19, 29):
                out.append(gaps("free N=%d s=%d" % (N, s), g, o32, o64))
                print(out[-1], flush=True)
        md, ma = g.checkSigma()
        out.append({"tag": "free N=%d end" % N, "min_diag": md, "max_asym": ma, "pos_err": maxabs(g.base_mu[:3], sc.pos)})
        print(out[-1], flush=True)
        g.close()

    # timing + profile at N=256
    N = 256
    sc = Scenario(N, seed=0)
    g = TightlyCoupledEKF(max_features=N, hooks=True)
    g.addNewFeatures(sc.initial_features())
    fr = list(sc.frames(40))
    g.upload_measurements(np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]), np.stack([f[2] for f in fr]))
    g.run_uploaded(0, 10, sc.dt)
    g.synchronize()
    t = time.time()
    g.run_uploaded(10, 30, sc.dt)
    g.synchronize()
    dtm = (time.time() - t) / 30
    md, ma = g.checkSigma()
    out.append({"tag": "N=256 loop", "ms_per_step": dtm * 1e3, "min_diag": md, "max_asym": ma,
                "pos_err": maxabs(g.base_mu[:3], sc.pos), "vel_err": maxabs(g.base_mu[7:10], sc.vel)})
    print(out[-1], flush=True)
    g.profile(True)
    g.run_uploaded(0, 10, sc.dt)
    g.synchronize()
    rep = g.profile_report()
    g.profile(False)
    out.append({"tag": "N=256 profile (10 steps)", "report": rep})
    print(json.dumps(rep, indent=1), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/gpu_diag.json", "w"), indent=1)


if __name__ == "__main__":
    main()
