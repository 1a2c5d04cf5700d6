"""Round 6: the N = 1024 teacher-forced step (tests/test_gpu_shapes.py::test_teacher_forced_one_step_n1024) with the second Joseph GEMM forming both
triangles (EKFVIO_SYM_JOSEPH=0, rounds 1-5) and the lower one mirrored (default): the error dictionary against the fp64 / fp32 oracle, from the
SAME starting state (made by the both-triangles flow), plus the free run from the raw prior.  GPU box: python scripts/sym_joseph_check.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF, capi  # noqa: E402
from ekf_vio_amd.sim import Scenario  # noqa: E402
from oracle import OracleFilter, max_threads, set_threads  # noqa: E402
from tests.test_gpu_shapes import _teacher_forced  # noqa: E402

set_threads(min(max_threads(), 16))
N = 1024
sc = Scenario(N, seed=0)
frames = list(sc.frames(4))
os.environ["EKFVIO_SYM_JOSEPH"] = "0"
g = TightlyCoupledEKF(max_features=N)
g.addNewFeatures(sc.initial_features())
st = g.get_state()
d = np.diag(st["Sigma"]).copy()
d[7:16] = 0.05
d[24::3] = 1.0
st["Sigma"] = np.diag(d).astype(np.float32)
st["base_mu"][7:10] = (-0.1, 0.0, -0.1)
st["base_mu"][10:13] = (0.0, 0.1, 0.0)
g.set_state(st)
for z, R, p in frames[:3]:
    g.process(sc.dt)
    assert g.updateWithFeaturePositions(z, R, p) == capi.OK
st32 = g.get_state()
g.close()
zz, RR, pp = frames[3]
pp = pp.copy()
pp[[5, 77, 500, 1023]] = 0
for sym in ("0", "1"):
    os.environ["EKFVIO_SYM_JOSEPH"] = sym
    g = TightlyCoupledEKF(max_features=N)
    o32, o64 = OracleFilter(np.float32), OracleFilter(np.float64)
    E, s64 = _teacher_forced(g, o32, o64, st32, sc.dt, zz, RR, pp)
    S = g.Sigma.astype(np.float64)
    lo = np.tril(np.ones(S.shape, dtype=bool))
    S64, S32 = s64["Sigma"].astype(np.float64), o32.get_state()["Sigma"].astype(np.float64)
    rl = lambda a, b, msk: float(np.linalg.norm((a - b)[msk]) / np.linalg.norm(b[msk]))  # noqa: E731
    sy = lambda M: 0.5 * (M + M.T)  # noqa: E731
    al = np.ones(S.shape, dtype=bool)
    print("   symmetric parts: gpu-64 %.3e o32-64 %.3e | antisymmetric part of the fp64 result %.3e of its norm, of the fp32 oracle's %.3e, of the input's %.3e" % (
        rl(sy(S), sy(S64), al), rl(sy(S32), sy(S64), al), np.linalg.norm(S64 - S64.T) / 2 / np.linalg.norm(S64),
        np.linalg.norm(S32 - S32.T) / 2 / np.linalg.norm(S32),
        np.linalg.norm(st32["Sigma"].astype(np.float64) - st32["Sigma"].astype(np.float64).T) / 2 / np.linalg.norm(st32["Sigma"].astype(np.float64))))
    print("sym=%s" % sym, {k: "%.3e" % v for k, v in E.items()},
          "lower: gpu-64 %.3e o32-64 %.3e gpu-o32 %.3e | upper: gpu-64 %.3e o32-64 %.3e gpu-o32 %.3e | asym gpu %.3e o32 %.3e" % (
              rl(S, S64, lo), rl(S32, S64, lo), rl(S, S32, lo), rl(S, S64, ~lo), rl(S32, S64, ~lo), rl(S, S32, ~lo),
              np.abs(S - S.T).max(), np.abs(S32 - S32.T).max()), flush=True)
    g.close(), o32.close(), o64.close()
