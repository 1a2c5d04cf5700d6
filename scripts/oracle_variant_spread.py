"""Tabulates tests/_variants.py: the spread between the Eigen builds the reference may have been (CPU only).
  python scripts/oracle_variant_spread.py > profiles/r05_oracle_variant_spread.txt"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import _variants as V  # noqa: E402

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
