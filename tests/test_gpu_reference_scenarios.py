"""The reference's OWN scenario inputs and deployment configurations as HIP-vs-oracle cases.

  * test/test_ekf.cpp:66-141 -- updateWithFeaturePositions straight from the diagonal prior (no process(dt) in front),
    R = 1e-3 I, z = the landmarks' own positions: 3 landmarks measured [T, F, T] (twice: the second call runs on the
    updated, no longer diagonal state), 103 landmarks with one unmeasured, 503 with 401 unmeasured (n = 1531, m = 204:
    n >> m in the persistent sweep) and 503 all measured (m = 1006, m_pad = 1024: sixteen block columns, the first size of
    the split sweep);
  * test/jacobian_test.cpp:50-72 -- F at 3 / 99 / 501 landmarks (the three points repeated), omega_x = 3.1415, b_dx = 1;
  * params/fast_with_insight.yaml and params/test.yaml -- num_features 400 / 30, default_point_depth_variance 1000,
    fast_threshold 45, inverse_image_scale 2: teacher-forced filter steps at both sizes with the yaml's prior, and the
    image loop at scale 2 with the yaml's detector settings.

Tolerances as written in tests/test_gpu_parity.py: bookkeeping, process(dt), tracker and detector bit-exact; update: HIP
forward error against the fp64 evaluation of the same step <= ACC_FACTOR x the fp32 oracle's + a floor.
"""
import numpy as np
import pytest

from ekf_vio_amd import EKFVIO, TightlyCoupledEKF, capi
from ekf_vio_amd.sim import Scenario, translated_sequence
from oracle import OracleFilter, max_threads, set_threads

from _oracle_node import OracleNode
from _scatter import backward_yardstick
from test_gpu_loop import K, grey
from test_gpu_shapes import ACC_FACTOR, MU_FLOOR, SIG_FLOOR, maxabs, relf, to32

pytestmark = pytest.mark.gpu
UV3 = [[0.1, 0.1], [-0.1, -0.1], [0.1, -0.1]]


@pytest.fixture()
def oracle_threads():
    set_threads(min(max_threads(), 16))
    yield
    set_threads(1)


def _compare_update(g, st, z, R, p, tag):
    """One update from the fp32 state `st` in HIP, oracle-fp32 and oracle-fp64; the yardstick assertion."""
    o32, o64 = OracleFilter(np.float32), OracleFilter(np.float64)
    g.set_state(st), o32.set_state(st), o64.set_state(st)
    rc = g.updateWithFeaturePositions(z, R, p)
    ro = o32.update(z, R, p)
    o64.update(z, R, p)
    sg, s32, s64 = g.get_state(), o32.get_state(), o64.get_state()
    assert rc == (capi.OK if ro == 0 else capi.ENUMERIC), (tag, rc, ro)
    assert np.array_equal(sg["del_flag"], s32["del_flag"]) and np.array_equal(sg["last_klt"], s32["last_klt"]), tag
    assert abs(np.linalg.norm(sg["base_mu"][3:7]) - 1) < 1e-6, tag
    E = dict(mu_gpu=maxabs(sg["base_mu"], s64["base_mu"]), mu_o32=maxabs(s32["base_mu"], s64["base_mu"]),
             feat_gpu=maxabs(sg["feat_mu"], s64["feat_mu"]), feat_o32=maxabs(s32["feat_mu"], s64["feat_mu"]),
             sig_gpu=relf(sg["Sigma"], s64["Sigma"]), sig_o32=relf(s32["Sigma"], s64["Sigma"]))
    assert E["mu_gpu"] <= ACC_FACTOR * E["mu_o32"] + MU_FLOOR, (tag, E)
    assert E["feat_gpu"] <= ACC_FACTOR * E["feat_o32"] + MU_FLOOR, (tag, E)
    assert E["sig_gpu"] <= ACC_FACTOR * E["sig_o32"] + SIG_FLOOR, (tag, E)
    return sg, E


def _ref_inputs(n_true_tail, n_false_tail, all_measured=False):
    """features / covs / measured exactly as test_ekf.cpp builds them: the three points, then (0.1, 0.1) repeated."""
    uv = np.array(UV3 + [[0.1, 0.1]] * (n_true_tail + n_false_tail), np.float32)
    R = np.tile(np.array([1e-3, 0, 0, 1e-3], np.float32), (len(uv), 1))
    p = np.array([1, 0, 1] + [1] * n_true_tail + [0] * n_false_tail, np.uint8)
    if all_measured:
        p[:] = 1
    return uv, R, p


@pytest.mark.parametrize("case", ["small", "medium", "large_with_false", "large_full"])
def test_reference_update_scenarios(oracle_threads, case):
    """test/test_ekf.cpp:66-141: a fresh filter, addNewFeatures, then the update -- from the DIAGONAL prior, no process(dt)."""
    tails = dict(small=(0, 0, False), medium=(100, 0, False), large_with_false=(100, 400, False), large_full=(100, 400, True))
    nt, nf, full = tails[case]
    uv, R, p = _ref_inputs(nt, nf, full)
    N = len(uv)
    g = TightlyCoupledEKF(max_features=N)
    g.addNewFeatures(uv)
    st = g.get_state()
    assert np.count_nonzero(st["Sigma"] - np.diag(np.diag(st["Sigma"]))) == 0  # the prior is diagonal
    m = 2 * int(p.sum())
    assert m == dict(small=4, medium=204, large_with_false=204, large_full=1006)[case]
    persistent0 = g.persistent_sweeps()
    sg, E = _compare_update(g, st, uv, R, p, case)
    # which sweep ran: one persistent launch from three block columns on, the split per-step sweep at sixteen
    took_persistent = g.persistent_sweeps() - persistent0
    assert took_persistent == (1 if case in ("medium", "large_with_false") else 0), (case, took_persistent)
    # unmeasured landmarks are flagged and keep their prior (nothing correlates them with the rest yet)
    un = np.flatnonzero(p == 0)
    assert np.array_equal(sg["del_flag"], (p == 0).astype(np.uint8))
    assert np.array_equal(sg["feat_mu"][un], st["feat_mu"][un])
    if case == "small":  # the reference calls the same update a second time on the updated state (:92-96)
        _compare_update(g, sg, uv, R, p, "small, second call")
    g.close()


@pytest.mark.parametrize("repeats", [1, 33, 167])  # 3, 99, 501 landmarks (jacobian_test.cpp:50-72)
def test_reference_jacobian_sizes_bit_exact(repeats):
    uv = np.array(UV3 * repeats, np.float32)
    g, o = TightlyCoupledEKF(max_features=len(uv)), OracleFilter(np.float32, emulate_static_cache=False)
    g.addNewFeatures(uv), o.add_new_features(uv)
    st = o.get_state()
    st["base_mu"][10], st["base_mu"][7] = 3.1415, 1.0
    o.set_state(st), g.set_state(st)
    for dt in (0.1, 0.0):
        Fg, Fo = g.numericallyLinearizeProcess(dt), o.linearize(dt)
        assert Fg.shape == (22 + 3 * len(uv),) * 2 and np.array_equal(Fg, Fo), (repeats, dt)
    g.close()


@pytest.mark.parametrize("N", [400, 30])  # params/fast_with_insight.yaml:2 and params/test.yaml:2
def test_yaml_configurations_teacher_forced(oracle_threads, N):
    """num_features 400 / 30 with default_point_depth_variance 1000 (both yaml files, :8): four teacher-forced filter steps
    behind a five-frame fp64 warm-up, incl. a frame with lost landmarks.  N = 400: n = 1222, m = 800 -> thirteen block
    columns, 407 owner tiles > 256 compute units -> the per-step sweep."""
    sc = Scenario(N, seed=3)
    kw = dict(default_point_depth_variance=1000.0)
    g = TightlyCoupledEKF(max_features=N, **kw)
    mk = lambda dt: OracleFilter(dt, depth_var=1000.0)
    o32, o64, teacher = mk(np.float32), mk(np.float64), mk(np.float64)
    uv = sc.initial_features()
    g.addNewFeatures(uv), teacher.add_new_features(uv)
    st0 = g.get_state()
    assert np.all(np.diag(st0["Sigma"])[24::3] == np.float32(1000.0))
    frames = list(sc.frames(9))
    for z, R, p in frames[:5]:
        teacher.process(sc.dt), teacher.update(z, R, p)
    fails = {2: max(N // 20, 1)}
    for s, (z, R, p) in enumerate(frames[5:]):
        p = p.copy()
        for q in range(fails.get(s, 0)):
            p[(7 * q + 3) % N] = 0
        st32 = to32(teacher.get_state())
        g.set_state(st32), o32.set_state(st32)
        g.process(sc.dt), o32.process(sc.dt)
        sg, so = g.get_state(), o32.get_state()
        for k in ("base_mu", "feat_mu", "Sigma"):
            assert np.array_equal(sg[k], so[k]), ("process", N, s, k)
        _compare_update(g, so, z, R, p, (N, s))
        teacher.process(sc.dt), teacher.update(z, R, p)
    g.close()


def test_yaml_first_update_from_the_depth_variance_1000_prior(oracle_threads):
    """The first update behind addNewFeatures with the yaml prior: cond(S) is ten times the default prior's.  One oracle run
    is no yardstick there (tests/test_gpu_parity.py): the backward-error yardstick of tests/_scatter.py applies."""
    N = 30
    sc = Scenario(N, seed=4)
    g = TightlyCoupledEKF(max_features=N, default_point_depth_variance=1000.0)
    o32, o64 = OracleFilter(np.float32, depth_var=1000.0), OracleFilter(np.float64, depth_var=1000.0)
    uv = sc.initial_features()
    g.addNewFeatures(uv), o32.add_new_features(uv)
    z, R, p = next(iter(sc.frames(1)))
    g.process(sc.dt), o32.process(sc.dt)
    st = o32.get_state()
    assert np.array_equal(g.get_state()["Sigma"], st["Sigma"])
    g.set_state(st), o64.set_state(st)
    rc = g.updateWithFeaturePositions(z, R, p)
    o32.update(z, R, p), o64.update(z, R, p)
    assert rc in (capi.OK, capi.ENUMERIC)
    sg, s32, s64 = g.get_state(), o32.get_state(), o64.get_state()
    yard = backward_yardstick(st, z, R, p, s64, c=8.0)
    assert maxabs(sg["base_mu"], s64["base_mu"]) <= max(yard["mu"], ACC_FACTOR * maxabs(s32["base_mu"], s64["base_mu"])) + MU_FLOOR
    assert relf(sg["Sigma"], s64["Sigma"]) <= yard["sig"] + ACC_FACTOR * relf(s32["Sigma"], s64["Sigma"]) + SIG_FLOOR
    g.close()


@pytest.mark.parametrize("N", [30, 400])
def test_yaml_image_loop_scale_2(N):
    """inverse_image_scale 2, fast_threshold 45, min_new_feature_dist 30, depth variance 1000 (both yaml files): six frames of
    the teacher-forced image loop through ekfvio_step_image with replenishment.  Pass flags, landmark counts, tracker results
    and the replenishment picks bit-exact against the oracle's node."""
    base = grey()
    seq = translated_sequence(base, 6, dx=-2.6, dy=-1.2)
    kw = dict(inverse_image_scale=2, fast_threshold=45, min_new_feature_dist=30, default_point_depth_variance=1000.0)
    v = EKFVIO(max_features=N, replenish=1, **kw)
    node = OracleNode(N, K, inverse_image_scale=2, fast_threshold=45, min_new_feature_dist=30, depth_var=1000.0)
    for i, img in enumerate(seq):
        stamp = 2.0 + i / 30.0
        if i > 0:
            v.tc_ekf.set_state(node.ekf.get_state())
        n_before = node.ekf.num_features
        rc = v.addFrame(stamp, img, K)
        node.add_frame(stamp, img)
        sg, so = v.tc_ekf.get_state(), node.ekf.get_state()
        assert rc in (capi.OK, capi.ENUMERIC)
        assert v.tc_ekf.num_features == node.ekf.num_features, i
        assert np.array_equal(sg["del_flag"], so["del_flag"]) and np.array_equal(sg["last_klt"], so["last_klt"]), i
        assert np.array_equal(sg["feat_mu"][n_before:], so["feat_mu"][n_before:]), i
        nn = 22 + 3 * n_before
        assert np.array_equal(sg["Sigma"][nn:, :], so["Sigma"][nn:, :]), i  # the new landmarks' rows: [hv, hv, 1000] on the diagonal
    assert node.ekf.num_features >= min(N, 20)
    v.tc_ekf.close()
