"""The spread between the x86-64 Eigen builds the reference may have been compiled as: SSE2 quaternion
product / reduction order (the oracle's and the HIP path's default), float sin/cos, `/=` as a reciprocal multiply -- on
the reference's own inputs, against the tolerances the HIP-vs-oracle tests state.  Numbers: scripts/oracle_variant_spread.py
-> profiles/r05_oracle_variant_spread.txt, DESIGN.md section 5."""
import numpy as np
import pytest

import _variants as V
from oracle import OracleFilter

F_TOL = 2e-4  # finite-difference Jacobian entries, abs (ulp(0.1 .. 1.5) / 2 delta)


def test_default_variant_is_the_sse2_build():
    """OracleFilter() with no switches == eigen_sse_quat=1, trig_float=0, div_reciprocal=0."""
    a, b = V.jacobian_case(1, 0.1, {}, generic=True), V.jacobian_case(1, 0.1, V.DEFAULT, generic=True)
    assert np.array_equal(a, b)
    c = V.jacobian_case(1, 0.1, dict(eigen_sse_quat=0), generic=True)
    assert not np.array_equal(a, c)  # the switch does something on a rotated state


def test_sse_quaternion_product_is_the_hamilton_product():
    """Both orders are the same product in exact arithmetic: fp64 evaluations agree to 1e-15, fp32 to a few ulp."""
    rng = np.random.default_rng(0)
    for _ in range(20):
        mu = np.zeros(22)
        q = rng.normal(size=4)
        mu[3:7] = q / np.linalg.norm(q)
        mu[7:16] = rng.normal(size=9)
        outs = []
        for sse in (1, 0):
            o = OracleFilter(np.float64, eigen_sse_quat=sse)
            outs.append(o.convolve_base_state(mu, 0.1))
            o.close()
        assert np.max(np.abs(outs[0] - outs[1])) < 1e-14
        o1, o0 = OracleFilter(np.float32, eigen_sse_quat=1), OracleFilter(np.float32, eigen_sse_quat=0)
        a, b = o1.convolve_base_state(mu, 0.1), o0.convolve_base_state(mu, 0.1)
        assert np.max(np.abs(a.astype(np.float64) - b)) < 4e-7
        assert np.max(np.abs(a - outs[0])) < 4e-7
        o1.close(), o0.close()


@pytest.mark.parametrize("repeats", [1, 33, 167])  # 3 / 99 / 501 landmarks, test/jacobian_test.cpp:50-72
@pytest.mark.parametrize("generic", [False, True])
def test_jacobian_variant_spread_is_inside_the_fd_tolerance(repeats, generic):
    if generic and repeats == 167:
        pytest.skip("same base block as the smaller sizes")
    for dt in (0.1, 0.0):
        r = V.jacobian_spread(repeats, dt, generic=generic)
        worst = max(r["spread"].values())
        assert worst <= F_TOL, (repeats, dt, r)
        assert r["fp32_vs_fp64"] <= 2 * F_TOL  # the default itself against its fp64 evaluation: the same scale


@pytest.mark.parametrize("N", [3, 103, 503])  # the sizes of test/test_ekf.cpp:66-141
def test_one_step_variant_spread_is_inside_the_update_yardstick(N):
    """process(dt) + update from a dense state: every variant lies closer to the default than the default lies to its own
    fp64 evaluation -- i.e. inside the yardstick (ACC_FACTOR x that gap) the HIP path is held to."""
    r = V.step_spread(N)
    g = r["fp32_vs_fp64"]
    for name, row in r["spread"].items():
        for phase in ("process", "update"):
            for k in ("mu", "feat", "sig"):
                floor = 2e-6 if k != "sig" else 2e-5
                assert row[phase][k] <= g[phase][k] + floor, (N, name, phase, k, row[phase][k], g[phase][k])


def test_simulation_variant_spread_99_steps():
    """test/analyzeEKFSimulation.cpp:233-244, scenario :244, 99 free-running steps from the raw prior: the variants
    decorrelate less than fp32 does from fp64, none raises a pivot warning, all track the truth alike."""
    r = V.simulation_spread()
    g = r["fp32_vs_fp64"]
    assert g["flagged"] == 0
    for name, row in r["spread"].items():
        assert row["flagged"] == 0, name
        assert row["mu"] <= g["mu"] and row["feat"] <= g["feat"] and row["sig"] <= g["sig"], (name, row, g)
        assert abs(row["pos_err"] - g["pos_err"]) < 1e-3, (name, row["pos_err"], g["pos_err"])
