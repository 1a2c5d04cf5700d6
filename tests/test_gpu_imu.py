"""SURVEY 8(f) F4, second half: the IMU measurement update behind cfg.use_imu, HIP against its CPU specification
(oracle imu_update, itself checked against numpy in tests/test_imu_oracle_cpu.py).  The reference only stubs the IMU callback
(EKFVIO.cpp:113-115): with use_imu = 0, the default, ekfvio_imu must change nothing."""
import numpy as np
import pytest

from ekf_vio_amd import EKFVIO, TightlyCoupledEKF, capi
from ekf_vio_amd.sim import Scenario
from oracle import OracleFilter

pytestmark = pytest.mark.gpu
G = (0.0, 9.81, 0.0)
ACC_FACTOR, MU_FLOOR, SIG_FLOOR = 4.0, 2e-5, 2e-6


def relf(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def rot_t(q, v):
    w, c = q[0], -np.asarray(q[1:], np.float64)
    uv = 2.0 * np.cross(c, v)
    return v + w * uv + np.cross(c, uv)


@pytest.mark.parametrize("N", [0, 30, 256])
def test_imu_update_teacher_forced_against_the_specification(N):
    sc = Scenario(max(N, 1), seed=7, dt=0.05)
    o64 = OracleFilter(np.float64)
    if N:
        o64.add_new_features(sc.initial_features()[:N])
        for z, R, p in sc.frames(5):
            o64.process(sc.dt), o64.update(z[:N], R[:N], p[:N])
    else:
        o64.process(0.05)
    st = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in o64.get_state().items()}
    g = TightlyCoupledEKF(max_features=max(N, 1), use_imu=1)
    o32 = OracleFilter(np.float32)
    worst = dict(mu_g=0.0, mu_o=0.0, f_g=0.0, f_o=0.0, s_g=0.0, s_o=0.0)
    rng = np.random.default_rng(N)
    for step in range(4):
        g.set_state(st), o32.set_state(st), o64.set_state(st)
        q = st["base_mu"][3:7].astype(np.float64)
        gyro = st["base_mu"][10:13] + st["base_mu"][19:22] + rng.normal(0, 1e-2, 3)
        acc = st["base_mu"][13:16] + st["base_mu"][16:19] - rot_t(q, np.array(G)) + rng.normal(0, 1e-1, 3)
        g.imuUpdate(gyro, acc)
        o32.imu_update(gyro.astype(np.float32), acc.astype(np.float32), 1e-4, 1e-2, G)
        o64.imu_update(gyro.astype(np.float32).astype(np.float64), acc.astype(np.float32).astype(np.float64),
                       float(np.float32(1e-4)), float(np.float32(1e-2)), np.array(G, np.float32).astype(np.float64))
        sg, s32, s64 = g.get_state(), o32.get_state(), o64.get_state()
        assert abs(np.linalg.norm(sg["base_mu"][3:7]) - 1) < 1e-6
        worst["mu_g"] = max(worst["mu_g"], np.abs(sg["base_mu"] - s64["base_mu"]).max())
        worst["mu_o"] = max(worst["mu_o"], np.abs(s32["base_mu"] - s64["base_mu"]).max())
        if N:
            worst["f_g"] = max(worst["f_g"], np.abs(sg["feat_mu"] - s64["feat_mu"]).max())
            worst["f_o"] = max(worst["f_o"], np.abs(s32["feat_mu"] - s64["feat_mu"]).max())
        worst["s_g"] = max(worst["s_g"], relf(sg["Sigma"], s64["Sigma"]))
        worst["s_o"] = max(worst["s_o"], relf(s32["Sigma"], s64["Sigma"]))
        st = s32  # the next step starts from the fp32 specification's result
        # a process step in between keeps the covariance dense and realistic
        o32.process(0.005)
        st = o32.get_state()
    assert worst["mu_g"] <= ACC_FACTOR * worst["mu_o"] + MU_FLOOR, worst
    assert worst["f_g"] <= ACC_FACTOR * worst["f_o"] + MU_FLOOR, worst
    assert worst["s_g"] <= ACC_FACTOR * worst["s_o"] + SIG_FLOOR, worst
    g.close()


def test_imu_callback_is_a_no_op_by_default_and_a_predict_plus_update_behind_the_flag():
    N = 20
    sc = Scenario(N, seed=1)
    fr = list(sc.frames(3))
    states = {}
    for flag in (0, 1):
        v = EKFVIO(max_features=N, use_imu=flag)
        e = v.tc_ekf
        e.addNewFeatures(sc.initial_features())
        for z, R, p in fr:
            e.process(sc.dt)
            e.updateWithFeaturePositions(z, R, p)
        before = e.get_state()
        v.imu_now(1.000, [0.0, 0.1, 0.0], [0.0, -9.81, 0.0])   # first record: only sets the filter's time
        mid = e.get_state()
        v.imu_now(1.005, [0.0, 0.1, 0.0], [0.0, -9.81, 0.0])   # 5 ms later: process(0.005) + update when enabled
        after = e.get_state()
        states[flag] = (before, mid, after)
        if flag:
            with pytest.raises(capi.EkfvioError) as ex:
                v.imu_now(0.9, [0, 0, 0], [0, 0, 0])           # a stamp before the filter's time
            assert ex.value.code == capi.EINVAL
        e.close()
    b0, m0, a0 = states[0]
    for k in ("base_mu", "feat_mu", "Sigma"):
        assert np.array_equal(b0[k], m0[k]) and np.array_equal(b0[k], a0[k]), k   # reference behaviour: nothing happens
    b1, m1, a1 = states[1]
    assert np.array_equal(b1["Sigma"], m1["Sigma"]) and not np.array_equal(m1["Sigma"], a1["Sigma"])
    # against the specification: process(0.005) then imu_update, from the same fp32 state
    o = OracleFilter(np.float32)
    o.set_state(m1)
    o.process(np.float32(1.005 - 1.000))
    o.imu_update(np.array([0.0, 0.1, 0.0], np.float32), np.array([0.0, -9.81, 0.0], np.float32), 1e-4, 1e-2, G)
    so = o.get_state()
    assert np.abs(a1["base_mu"] - so["base_mu"]).max() < 5e-5 and relf(a1["Sigma"], so["Sigma"]) < 2e-4
    # the gyro reading pulled the angular rate towards it and made it better known
    assert abs(a1["base_mu"][11] + a1["base_mu"][20] - 0.1) < abs(m1["base_mu"][11] + m1["base_mu"][20] - 0.1) + 1e-6
    assert a1["Sigma"][11, 11] < m1["Sigma"][11, 11]


def test_imu_records_between_frames_keep_the_closed_loop_on_track():
    """IMU at 200 Hz between camera frames at 25 Hz (eight 5 ms ticks per frame; BASELINE config 2 names 200 Hz / 30 Hz):
    per tick one propagate + IMU update from consistent synthetic readings (the truth's rate and specific force at that
    tick), on every eighth tick also the camera update.  The filter must track the truth as it does without the IMU."""
    N, frames, tick = 64, 30, 0.005
    err = {}
    for flag in (0, 1):
        truth = Scenario(N, seed=3, dt=tick)
        v = EKFVIO(max_features=N, use_imu=flag, gravity=G)
        e = v.tc_ekf
        e.addNewFeatures(truth.initial_features())
        t = 0.0
        v.imu_now(t, truth.omega, truth.acc - rot_t(truth.quat, np.array(G)))  # sets the filter's clock
        for _ in range(frames):
            for k in range(8):
                truth.advance()
                t += tick
                if flag:
                    v.imu_now(t, truth.omega, truth.acc - rot_t(truth.quat, np.array(G)))
            if not flag:
                e.process(np.float32(8 * tick))
            z, R, p = truth.measure()
            assert e.updateWithFeaturePositions(z, R, p) in (capi.OK, capi.ENUMERIC)
        b = e.base_mu
        err[flag] = (np.abs(b[:3] - truth.pos).max(), np.abs(b[10:13] - truth.omega).max(), np.abs(b[7:10] - truth.vel).max())
        assert np.isfinite(e.Sigma).all() and e.checkSigma()[0] >= 0
        e.close()
    print("closed loop, (position, rate, velocity) error without / with IMU:", err)
    assert err[0][0] < 0.02 and err[1][0] < 0.02, err
    assert err[1][1] < 5e-3 and err[1][2] < 0.02, err
