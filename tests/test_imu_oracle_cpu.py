"""SURVEY 8(f) F4, second half: the IMU measurement update.  The reference only stubs it (EKFVIO.cpp:113-115), so the
specification is this repository's own (oracle/ekf_oracle.hpp: imu_update); here it is checked against an independent numpy
evaluation of the same model -- H by central differences of h(x), K = Sigma H^T S^-1 by numpy, Joseph form -- and against
the properties an EKF update must have."""
import numpy as np

from ekf_vio_amd.sim import Scenario
from oracle import OracleFilter

G = np.array([0.0, 9.81, 0.0])


def rot_t(q, v):
    """R(q)^T v with the filter's rotation formula on the conjugate (q need not be normalised)."""
    w, c = q[0], -np.asarray(q[1:])
    uv = 2.0 * np.cross(c, v)
    return v + w * uv + np.cross(c, uv)


def h_imu(base):
    return np.concatenate([base[10:13] + base[19:22], base[13:16] + base[16:19] - rot_t(base[3:7], G)])


def converged_state(N=12, steps=6, seed=2):
    sc = Scenario(N, seed=seed, dt=0.05)
    o = OracleFilter(np.float64)
    o.add_new_features(sc.initial_features())
    for z, R, p in sc.frames(steps):
        o.process(sc.dt), o.update(z, R, p)
    return o


def test_rt_gravity_and_its_jacobian():
    o = OracleFilter(np.float64)
    rng = np.random.default_rng(0)
    for _ in range(5):
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q) * rng.uniform(0.9, 1.1)  # not exactly unit
        out, jac = o.rt_gravity(q, G)
        assert np.allclose(out, rot_t(q, G), rtol=0, atol=1e-13)
        num = np.zeros((3, 4))
        for k in range(4):
            d = np.zeros(4)
            d[k] = 1e-6
            num[:, k] = (rot_t(q + d, G) - rot_t(q - d, G)) / 2e-6
        assert np.allclose(jac, num, rtol=0, atol=1e-8)
    # unit quaternion: a proper rotation (norm preserved)
    q = np.array([np.cos(0.3), 0.0, np.sin(0.3), 0.0])
    assert abs(np.linalg.norm(o.rt_gravity(q, G)[0]) - 9.81) < 1e-12


def test_imu_update_matches_independent_numpy_evaluation():
    o = converged_state()
    st = o.get_state()
    n = o.dim
    base, Sig = st["base_mu"].copy(), st["Sigma"].copy()
    gyro = np.array([0.01, 0.12, -0.02])
    acc = np.array([0.05, -9.7, 0.1])
    gv, av = 1e-4, 1e-2
    o.imu_update(gyro, acc, gv, av, G)
    out = o.get_state()
    # independent evaluation
    H = np.zeros((6, n))
    for k in range(22):
        d = np.zeros(22)
        d[k] = 1e-6
        H[:, k] = (h_imu(base + d) - h_imu(base - d)) / 2e-6
    R = np.diag([gv] * 3 + [av] * 3)
    S = H @ Sig @ H.T + R
    K = Sig @ H.T @ np.linalg.inv(S)
    I_KH = np.eye(n) - K @ H
    Sig2 = I_KH @ Sig @ I_KH.T + K @ R @ K.T
    mu = np.concatenate([base, st["feat_mu"].ravel()]) + K @ (np.concatenate([gyro, acc]) - h_imu(base))
    mu[3:7] /= np.linalg.norm(mu[3:7])
    assert np.allclose(out["base_mu"], mu[:22], rtol=0, atol=1e-9)
    assert np.allclose(out["feat_mu"].ravel(), mu[22:], rtol=0, atol=1e-9)
    assert np.linalg.norm(out["Sigma"] - Sig2) / np.linalg.norm(Sig2) < 1e-8
    # an update never increases a variance, and it does inform the measured states
    assert (np.diag(out["Sigma"]) <= np.diag(Sig) * (1 + 1e-12) + 1e-15).all()
    post = np.diag(H @ out["Sigma"] @ H.T)
    assert (post < np.diag(R)).all()  # the measured combinations end up better known than one reading alone


def test_imu_update_fp32_close_to_fp64():
    o64 = converged_state()
    st = o64.get_state()
    o32 = OracleFilter(np.float32)
    o32.set_state({k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in st.items()})
    o64.set_state(o32.get_state())
    gyro, acc = [0.0, 0.1, 0.0], [0.0, -9.81, 0.0]
    o32.imu_update(gyro, acc, 1e-4, 1e-2, G), o64.imu_update(gyro, acc, 1e-4, 1e-2, G)
    a, b = o32.get_state(), o64.get_state()
    assert np.abs(a["base_mu"] - b["base_mu"]).max() < 1e-5
    assert np.linalg.norm(a["Sigma"] - b["Sigma"]) / np.linalg.norm(b["Sigma"]) < 1e-5
