"""Tabulates tests/_variants.py: the spread between the Eigen builds the reference may have been (CPU only).
  python scripts/oracle_variant_spread.py > profiles/r06_oracle_variant_spread.txt"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np  # noqa: E402
import _variants as V  # noqa: E402
from ekf_vio_amd.sim import Scenario  # noqa: E402
from oracle import OracleFilter, amd_order  # noqa: E402

print("# spread between oracle variants (fp32), each against the default sse=1 trigf=0 recip=0; 'fp32|fp64' = the default")
print("# against its own fp64 evaluation, the gap that sizes the HIP-vs-oracle tolerances.  host libm: this box's glibc.")
print("\n## finite-difference Jacobian F, max |dF| (test/jacobian_test.cpp:34-72 inputs; generic = rotated moving base state)")
for generic in (False, True):
    for rep in (1, 33, 167):
        for dt in (0.1, 0.0):
            r = V.jacobian_spread(rep, dt, generic=generic)
            print("N=%3d dt=%.1f %s  fp32|fp64 %.3g   " % (3 * rep, dt, "generic " if generic else "jac_test", r["fp32_vs_fp64"])
                  + "  ".join("%s: %.3g" % kv for kv in r["spread"].items()))
print("\n## one process(dt) + update from a dense state (sizes of test/test_ekf.cpp:66-141): base mean abs / landmark abs / Sigma rel-Frobenius")
for N in (3, 103, 503):
    r = V.step_spread(N)
    for phase in ("process", "update"):
        g = r["fp32_vs_fp64"][phase]
        print("N=%3d %-7s fp32|fp64        mu %.3g feat %.3g sig %.3g" % (N, phase, g["mu"], g["feat"], g["sig"]))
        for name, row in r["spread"].items():
            q = row[phase]
            print("N=%3d %-7s %s mu %.3g feat %.3g sig %.3g" % (N, phase, name, q["mu"], q["feat"], q["sig"]))
print("\n## 99 free-running steps, N=30, dt=0.05, scenario test/analyzeEKFSimulation.cpp:244")
r = V.simulation_spread()
g = r["fp32_vs_fp64"]
print("fp32|fp64              mu %.3g feat %.3g sig %.3g  flagged %d/%d  position error vs truth %.4g (fp64 %.4g)"
      % (g["mu"], g["feat"], g["sig"], g["flagged"], g["flagged64"], g["pos_err"], g["pos_err64"]))
for name, row in r["spread"].items():
    print("%s mu %.3g feat %.3g sig %.3g  flagged %d  position error vs truth %.4g" % (name, row["mu"], row["feat"], row["sig"], row["flagged"], row["pos_err"]))

print("\n## round 6: SimplicialLDLT's default ordering (AMDOrdering, TightlyCoupledEKF.cpp:577; oracle/amd_order.hpp), switch ldlt_amd_order (default 1)")
print("# complete graphs (a numerically dense S): identity for every size, with and without the diagonal in the pattern:",
      all(np.array_equal(amd_order(np.ones((n, n), bool), kd), np.arange(n)) for n in (2, 4, 6, 60, 100, 101, 102, 103, 204, 512, 1006) for kd in (True, False)))
print("# 2 x 2 block-diagonal patterns (update straight from the diagonal prior, test/test_ekf.cpp:66-141; m = 4, 204, 1006): identity:",
      all(np.array_equal(amd_order(np.kron(np.eye(b, dtype=bool), np.ones((2, 2), bool)), kd), np.arange(2 * b)) for b in (2, 102, 503) for kd in (True, False)))
for N, steps in ((3, 6), (30, 99), (103, 4)):
    sc = Scenario(N, seed=0, dt=0.05)
    o = OracleFilter(np.float32)
    o.add_new_features(sc.initial_features())
    nat = 0
    for z, R, p in sc.frames(steps):
        o.process(sc.dt)
        o.update(z, R, p)
        nat += int(np.array_equal(o.last_perm(), np.arange(2 * N)))
    o.close()
    print("# free-running N=%d, %d updates (m = %d): the ordering was the identity in %d of them -> ldlt_amd_order 1 | 0 give the same bits there" % (N, steps, 2 * N, nat))
# a pattern that does permute (built by hand: landmark 0's u coordinate correlated with every other landmark's, nothing else)
N = 10
sc = Scenario(N, seed=7)
uv = sc.initial_features()
z, R, p = list(sc.frames(1))[0]
res = {}
for name, dt, kw in (("amd", np.float32, {}), ("amd, diagonal dropped", np.float32, dict(amd_keep_diagonal=0)), ("natural", np.float32, dict(ldlt_amd_order=0)), ("fp64", np.float64, {})):
    o = OracleFilter(dt, **kw)
    o.add_new_features(uv)
    st = o.get_state()
    sig = st["Sigma"].copy()
    for i in range(1, N):
        sig[22, 22 + 3 * i] = sig[22 + 3 * i, 22] = 5e-7
    o.set_state({**st, "Sigma": sig})
    o.update(z, R, p)
    res[name] = (o.get_state(), o.last_perm().tolist())
    o.close()
print("# hand-built hub pattern, N=10 (m = 20): P =", res["amd"][1])
for name in ("amd", "amd, diagonal dropped"):
    a, b, c = res[name][0], res["natural"][0], res["fp64"][0]
    print("%-22s vs natural order: mu %.3g feat %.3g sig %.3g   (natural fp32|fp64: mu %.3g feat %.3g sig %.3g)" % (
        name, V.maxabs(a["base_mu"], b["base_mu"]), V.maxabs(a["feat_mu"], b["feat_mu"]), V.relf(a["Sigma"], b["Sigma"]),
        V.maxabs(b["base_mu"], c["base_mu"]), V.maxabs(b["feat_mu"], c["feat_mu"]), V.relf(b["Sigma"], c["Sigma"])))
