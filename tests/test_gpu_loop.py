"""The whole addFrame sequence over several frames against the oracle's sequence.

  * addNewFeatures on a DENSE covariance (after updates, with the P / P2 ping-pong behind it), through
    ekfvio_add_features and through a replenishing ekfvio_step_image (device-side count);
  * a teacher-forced image loop: every frame starts from the oracle's state, runs ekfvio_step_image(cfg.replenish = 1)
    and is compared with process -> klt_track -> update -> replenish of the oracle (tests/_oracle_node.py);
  * the free-running loop: landmarks lost / numeric warnings, HIP against the oracle loop on the same sequence;
  * frames larger than 640x480 (752x480, 1280x960; scale 1, 2, 4): pyramid, tracker, FAST, replenishment bit-exact
    (the global occupancy mask of replenish_select_kernel<false>, pyramids beyond 20x15 tiles);
  * a handle on device 1 where the box has one.

Reference: EKFVIO.cpp:139-196,224-311, TightlyCoupledEKF.cpp:58-94.  Tolerances as written in tests/test_gpu_parity.py:
bookkeeping, process(dt), tracker and detector bit-exact; update within ACC_FACTOR x the fp32 oracle's error against
the fp64 evaluation of the same step + a floor.
"""
import os

import numpy as np
import pytest
from PIL import Image

from ekf_vio_amd import EKFVIO, KLTTracker, TightlyCoupledEKF, capi
from ekf_vio_amd.sim import Scenario, translated_sequence
from oracle import KltFrame, OracleFilter, fast_detect, frame_resize, klt_track, replenish

from _oracle_node import OracleNode
from _scatter import backward_yardstick
from test_gpu_shapes import ACC_FACTOR, MU_FLOOR, SIG_FLOOR, maxabs, relf

pytestmark = pytest.mark.gpu
IMG = os.path.join(os.path.dirname(__file__), "golden", "images")
K = np.array([500.0, 0, 320.0, 0, 500.0, 240.0, 0, 0, 1.0], np.float32)
KEYS = ("base_mu", "feat_mu", "last_klt", "del_flag", "Sigma")


def grey(name="640_480_test"):
    return np.asarray(Image.open(os.path.join(IMG, name + "_gray.png")))


def big_image(w, h):
    """A textured frame larger than the reference's test image: the 640x480 image tiled with mirrored copies (no seams
    that FAST would fire on along straight lines only) and a smooth brightness ramp so tiles are not identical."""
    base = grey().astype(np.int32)
    row = np.concatenate([base, base[:, ::-1], base], axis=1)
    full = np.concatenate([row, row[::-1], row], axis=0)[:h, :w]
    yy, xx = np.mgrid[0:h, 0:w]
    ramp = ((xx * 13) // w + (yy * 7) // h) - 10
    return np.ascontiguousarray(np.clip(full + ramp, 0, 255).astype(np.uint8))


# ------------------------------------------------------------------ (a) addNewFeatures on a dense covariance
def test_add_features_on_dense_sigma_host_count():
    """Three steps at N = 30 of capacity 64 make Sigma dense and leave the live covariance in the ping-pong partner;
    then addNewFeatures(10): Sigma, mu, last_klt and the flags must be the oracle's bit for bit (new rows and columns
    zero, new diagonal [hv, hv, dv], TightlyCoupledEKF.cpp:58-94), and the next teacher-forced step must agree."""
    N0, cap, add = 30, 64, 10
    sc = Scenario(N0 + add, seed=5)
    uv_all = sc.initial_features()
    g = TightlyCoupledEKF(max_features=cap)
    o32, o64 = OracleFilter(np.float32), OracleFilter(np.float64)
    g.addNewFeatures(uv_all[:N0])
    frames = list(sc.frames(4))
    for z, R, p in frames[:3]:
        g.process(sc.dt)
        assert g.updateWithFeaturePositions(z[:N0], R[:N0], p[:N0]) in (capi.OK, capi.ENUMERIC)
    st = g.get_state()
    assert np.count_nonzero(st["Sigma"]) > 0.8 * st["Sigma"].size
    o32.set_state(st)
    g.addNewFeatures(uv_all[N0:]), o32.add_new_features(uv_all[N0:])
    sg, so = g.get_state(), o32.get_state()
    assert g.num_features == N0 + add and g.dim == 22 + 3 * (N0 + add)
    for k in KEYS:
        assert np.array_equal(sg[k], so[k]), k
    n0 = 22 + 3 * N0
    assert np.array_equal(sg["Sigma"][:n0, :n0], st["Sigma"]) and not sg["Sigma"][n0:, :n0].any() and not sg["Sigma"][:n0, n0:].any()
    # one more step from that state: predict bit-exact, update inside the yardstick
    z, R, p = frames[3]
    g.process(sc.dt), o32.process(sc.dt)
    sg, so = g.get_state(), o32.get_state()
    for k in ("base_mu", "feat_mu", "Sigma"):
        assert np.array_equal(sg[k], so[k]), ("process after growth", k)
    o64.set_state(so)
    assert g.updateWithFeaturePositions(z, R, p) in (capi.OK, capi.ENUMERIC)
    o32.update(z, R, p), o64.update(z, R, p)
    sg, s32, s64 = g.get_state(), o32.get_state(), o64.get_state()
    assert np.array_equal(sg["del_flag"], s32["del_flag"]) and np.array_equal(sg["last_klt"], s32["last_klt"])
    # the ten new landmarks come with the raw prior (variance 100 on inverse depth): this update is as ill-conditioned as
    # a first one, so it is held to the backward-error yardstick of tests/_scatter.py
    yard = backward_yardstick(so, z, R, p, s64, c=8.0)
    assert maxabs(sg["base_mu"], s64["base_mu"]) <= max(yard["mu"], ACC_FACTOR * maxabs(s32["base_mu"], s64["base_mu"])) + MU_FLOOR
    assert relf(sg["Sigma"], s64["Sigma"]) <= yard["sig"] + ACC_FACTOR * relf(s32["Sigma"], s64["Sigma"]) + SIG_FLOOR
    g.close()


def test_add_features_on_dense_sigma_device_count():
    """The same growth through a replenishing ekfvio_step_image: the number of new landmarks stays on the device
    (add_features_enqueue_device_count) and Sigma grows behind the update inside the frame.  Frame k of a translated
    sequence loses landmarks to the kill box, so frames 2.. replenish onto a dense covariance."""
    base = grey()
    seq = translated_sequence(base, 5, dx=-6.0, dy=-2.5)  # fast enough that landmarks leave through the kill box
    v = EKFVIO(max_features=160, replenish=1)  # more than one 640x480 frame yields at 30 px spacing: every frame may add
    node = OracleNode(160, K)
    grew = 0
    for i, img in enumerate(seq):
        stamp = 2.0 + i / 30.0
        if i > 0:
            v.tc_ekf.set_state(node.ekf.get_state())  # teacher forcing: both sides start the frame from the same state
        n_before = node.ekf.num_features
        rc = v.addFrame(stamp, img, K)
        assert rc in (capi.OK, capi.ENUMERIC)
        node.add_frame(stamp, img)
        if i > 0:
            # a landmark that failed stays in the state (flagged), so the count only grows when max_features allows
            pre = node.last["pre_update"]
            o64 = OracleFilter(np.float64)
            o64.set_state(pre)
            o64.update(node.last["z"], node.last["R"], node.last["passed"])
        sg, so = v.tc_ekf.get_state(), node.ekf.get_state()
        assert v.tc_ekf.num_features == node.ekf.num_features, i
        k_new = node.ekf.num_features - n_before
        grew += (i > 0 and k_new > 0)
        assert np.array_equal(sg["del_flag"], so["del_flag"]) and np.array_equal(sg["last_klt"], so["last_klt"]), i
        n_old = 22 + 3 * n_before
        # the new landmarks and their rows / columns / diagonal: bit-exact whatever the update's rounding
        assert np.array_equal(sg["feat_mu"][n_before:], so["feat_mu"][n_before:]), i
        assert np.array_equal(sg["Sigma"][n_old:, :], so["Sigma"][n_old:, :]) and np.array_equal(sg["Sigma"][:, n_old:], so["Sigma"][:, n_old:]), i
    assert grew >= 1, "no frame replenished onto a dense covariance: the test does not exercise what it is named for"
    v.tc_ekf.close()


# ------------------------------------------------------------------ (b) teacher-forced and free-running image loops
def test_teacher_forced_image_loop_with_replenishment():
    """Ten frames of translated_sequence through ekfvio_step_image(cfg.replenish = 1).  Per frame, from the oracle's
    state: pass flags, landmark count, new landmarks' pixels and last_klt bit-exact; the state within the yardstick."""
    base = grey()
    seq = translated_sequence(base, 10, dx=-3.1, dy=-1.3)
    v = EKFVIO(max_features=160, replenish=1)  # never full: the replenishment adds landmarks on later frames too
    node = OracleNode(160, K)
    o64 = OracleFilter(np.float64)
    updates = grown = 0
    for i, img in enumerate(seq):
        stamp = 7.0 + i / 30.0
        if i > 0:
            v.tc_ekf.set_state(node.ekf.get_state())
        n_before = node.ekf.num_features
        rc_g = v.addFrame(stamp, img, K)
        rc_o = node.add_frame(stamp, img)
        sg, so = v.tc_ekf.get_state(), node.ekf.get_state()
        assert v.tc_ekf.num_features == node.ekf.num_features, i
        assert np.array_equal(sg["del_flag"], so["del_flag"]), i      # pass flags (a failed landmark is flagged, :528)
        assert np.array_equal(sg["last_klt"], so["last_klt"]), i      # tracker results and the new landmarks' positions
        assert np.array_equal(sg["feat_mu"][n_before:], so["feat_mu"][n_before:]), i  # = the new landmarks' pixels
        # what the node publishes after addFrame arrives with the frame's status word (frame_outputs_kernel): it must be the
        # state's own numbers, new landmarks included (their count is only known on the device when the kernel runs)
        od = v.odometry()
        xyz, inten = v.points()
        assert np.array_equal(od["position"], sg["base_mu"][0:3]) and np.array_equal(od["orientation_wxyz"], sg["base_mu"][3:7])
        zinv = (1.0 / sg["feat_mu"][:, 2].astype(np.float64)).astype(np.float32)
        assert xyz.shape == (v.tc_ekf.num_features, 3)
        assert np.array_equal(xyz, np.stack([sg["feat_mu"][:, 0] * zinv, sg["feat_mu"][:, 1] * zinv, zinv], axis=1)), i
        if i == 0:
            assert len(node.last["new_px"]) > 10
            for k in KEYS:
                assert np.array_equal(sg[k], so[k]), k
            continue
        updates += 1
        grown += len(node.last["new_px"]) > 0
        pre = node.last["pre_update"]
        o64.set_state(pre)
        o64.update(node.last["z"], node.last["R"], node.last["passed"])
        s64 = o64.get_state()
        nb = 22 + 3 * n_before
        e_g, e_o = maxabs(sg["base_mu"], s64["base_mu"]), maxabs(so["base_mu"], s64["base_mu"])
        f_g, f_o = maxabs(sg["feat_mu"][:n_before], s64["feat_mu"]), maxabs(so["feat_mu"][:n_before], s64["feat_mu"])
        r_g, r_o = relf(sg["Sigma"][:nb, :nb], s64["Sigma"]), relf(so["Sigma"][:nb, :nb], s64["Sigma"])
        # the tracker's R = 1e-5 px^2 / fx^2 = 4e-11 makes every update of this loop a cond(S) ~ 1e10..1e12 problem:
        # held to the backward-error yardstick (tests/_scatter.py), like the first update from the raw prior
        yard = backward_yardstick(pre, node.last["z"], node.last["R"], node.last["passed"], s64, c=8.0)
        assert e_g <= max(yard["mu"], ACC_FACTOR * e_o) + MU_FLOOR, (i, e_g, e_o, yard)
        assert f_g <= max(yard.get("feat", yard["mu"]), ACC_FACTOR * f_o) + 10 * MU_FLOOR, (i, f_g, f_o, yard)
        assert r_g <= yard["sig"] + ACC_FACTOR * r_o + SIG_FLOOR, (i, r_g, r_o, yard)
        if rc_o == 0:
            assert rc_g == capi.OK, i
    assert updates == 9 and grown >= 2
    v.tc_ekf.close()


def test_frame_outputs_published_between_the_joseph_gemms_are_the_updated_state(monkeypatch):
    """A frame that adds no landmarks publishes its outputs and its status between the update's two Joseph GEMMs (round 4: the host's
    next frame starts while the second GEMM runs): frame_outputs_kernel forms mu + K y (quaternion renormalised) itself, from K y in column n
    of P, where the second GEMM's first workgroup will find it.  The outputs must be the updated state's own numbers, and the whole
    sequence must agree bit for bit with EKFVIO_EARLY_OUTPUTS=0 (outputs behind the last kernel)."""
    import ctypes as C
    base = grey()
    seq = translated_sequence(base, 14, dx=-1.4, dy=-0.45)
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("EKFVIO_EARLY_OUTPUTS", mode)
        v = EKFVIO(max_features=48, replenish=1)  # full after the first frame: later frames add nothing
        rows = []
        for i, img in enumerate(seq):
            v.addFrame(7.0 + i / 30.0, img, K)
            od = v.odometry()
            xyz, inten = v.points()
            sg = v.tc_ekf.get_state()
            assert np.array_equal(od["position"], sg["base_mu"][0:3]) and np.array_equal(od["orientation_wxyz"], sg["base_mu"][3:7]), (mode, i)
            zinv = (1.0 / sg["feat_mu"][:, 2].astype(np.float64)).astype(np.float32)
            assert np.array_equal(xyz, np.stack([sg["feat_mu"][:, 0] * zinv, sg["feat_mu"][:, 1] * zinv, zinv], axis=1)), (mode, i)
            rows.append((od["position"].copy(), od["orientation_wxyz"].copy(), xyz.copy(), inten.copy(), sg["Sigma"].copy()))
        # (the C entry point the Python mirror no longer goes through: the same four slices of the base state)
        p3, q4, l3, a3 = (np.zeros(k, np.float32) for k in (3, 4, 3, 3))
        fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
        assert v.tc_ekf.lib.ekfvio_get_odometry(v.tc_ekf.h, fp(p3), fp(q4), fp(l3), fp(a3)) == 0
        od = v.odometry()
        assert np.array_equal(p3, od["position"]) and np.array_equal(q4, od["orientation_wxyz"])
        assert np.array_equal(l3, od["linear"]) and np.array_equal(a3, od["angular"])
        n_early = v.tc_ekf.counters()["early_output_frames"]
        if mode == "1":
            assert n_early >= 8, n_early  # (frames with all 48 landmarks alive; a lost landmark makes the next frame replenish)
        else:
            assert n_early == 0
        runs[mode] = rows
        v.tc_ekf.close()
    for i, (a, b) in enumerate(zip(runs["1"], runs["0"])):
        for x, y in zip(a, b):
            assert np.array_equal(x, y), i


@pytest.mark.parametrize("thr,dist,expect_n", [(50, 30, 90), (20, 12, 250)])
def test_free_running_loop_loses_what_the_oracle_loop_loses(thr, dist, expect_n):
    """46 frames of a 1.4 px / frame translation, capacity 256.  With the node's default
    detector (FAST 50, 30 px) the 640x480 image yields ~105 landmarks; with FAST 20 / 12 px -- the settings of bench.py's
    `full_loop.n256` -- the filter really runs 256.  Both loops run free (no teacher forcing); their rounding differs, so
    states are compared loosely -- what must agree is the behaviour: how many landmarks are ever lost, how many frames raise
    the numeric warning, and that both track the image motion."""
    base = grey()
    frames = 46
    seq = translated_sequence(base, frames)
    v = EKFVIO(max_features=256, replenish=1, fast_threshold=thr, min_new_feature_dist=dist)
    node = OracleNode(256, K, fast_threshold=thr, min_new_feature_dist=dist)
    warn_g = warn_o = 0
    for i, img in enumerate(seq):
        stamp = 1.0 + i / 30.0
        rc = v.addFrame(stamp, img, K)
        warn_g += rc == capi.ENUMERIC
        ro = node.add_frame(stamp, img)
        warn_o += ro == 1
        if i == 0:
            assert v.tc_ekf.num_features == node.ekf.num_features
    sg, so = v.tc_ekf.get_state(), node.ekf.get_state()
    lost_g, lost_o = int(sg["del_flag"].sum()), int(so["del_flag"].sum())
    n_g, n_o = v.tc_ekf.num_features, node.ekf.num_features
    line = "free-running 46 frames, FAST %d / %d px: HIP lost %d of %d (warnings %d), oracle loop lost %d of %d (warnings %d)" % (
        thr, dist, lost_g, n_g, warn_g, lost_o, n_o, warn_o)
    print(line)
    os.makedirs(os.path.join(os.path.dirname(__file__), "..", "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "free_running_loop.txt"), "a") as fh:
        fh.write(line + "\n")
    assert n_o >= expect_n and n_g >= expect_n, (n_g, n_o)  # the case really runs the landmark count it is named for
    assert np.isfinite(sg["base_mu"]).all() and np.isfinite(sg["Sigma"]).all()
    # the HIP loop may not be worse than the reference arithmetic by more than a few landmarks / frames
    assert lost_g <= lost_o + max(4, lost_o // 4), (lost_g, lost_o)
    assert warn_g <= warn_o + 3, (warn_g, warn_o)
    assert abs(n_g - n_o) <= 8
    # both see the same image motion on the landmarks neither lost (first batch, same detector picks)
    n1 = min(n_g, n_o)
    both = (sg["del_flag"][:n1] == 0) & (so["del_flag"][:n1] == 0)
    same_start = both.copy()
    assert both.sum() > 0.7 * n1
    d = (sg["last_klt"][:n1][same_start] - so["last_klt"][:n1][same_start]) * np.array([K[0], K[4]], np.float32)
    assert np.median(np.abs(d)) < 0.05  # pixels
    v.tc_ekf.close()


# ------------------------------------------------------------------ (c) frames larger than 640x480
@pytest.mark.parametrize("w,h,scale", [(752, 480, 1), (752, 480, 2), (1280, 960, 1), (1280, 960, 2), (1280, 960, 4)])
def test_large_frames_pyramid_tracker_fast_replenish_bit_exact(w, h, scale):
    """Frame::Frame takes any image (Frame.cpp:15-42).  752x480 (MT9V034) and 1280x960 cameras at scale 1, 2, 4: every
    pyramid level, the Scharr derivatives, tracked positions, FAST keypoints and the replenishment picks are the
    oracle's, with no size in the configuration (the device planes grow on the first frame that needs it)."""
    a = big_image(w, h)
    b = np.ascontiguousarray(np.roll(a, (3, -5), axis=(0, 1)))
    Kc = np.array([700.0, 0, w / 2, 0, 700.0, h / 2, 0, 0, 1.0], np.float32)
    v = EKFVIO(max_features=200, inverse_image_scale=scale)  # default max_image_width / height: 640 x 480
    t = v.tracker
    t.push_frame(a, Kc)
    ra = frame_resize(a, scale) if scale > 1 else a
    rb = frame_resize(b, scale) if scale > 1 else b
    oa = KltFrame(ra)
    for l in range(oa.levels):
        gi, gd = t.level(l)
        oi, od = oa.level(l)
        assert np.array_equal(gi, oi), ("image", l)
        assert np.array_equal(gd, od), ("deriv", l)
    # FAST + first fit on the first frame (ceil(w / 32) * h > 16384 words at 1280x960 scale 1: the global occupancy mask)
    xy, sc = v.fast(40, True)
    rxy, rsc = fast_detect(ra, 40, True)
    assert len(rxy) > 50 and np.array_equal(xy, rxy) and np.array_equal(sc, rsc)
    px = v.replenishFeatures()
    ref = replenish(ra, np.zeros((0, 2), np.float32), 200)
    assert len(ref) > 20 and np.array_equal(px, ref)
    # second frame: tracker on arbitrary points incl. the far corner region
    t.push_frame(b, Kc)
    ob = KltFrame(rb)
    hh, ww = ra.shape
    pts = np.array([[x, y] for y in np.linspace(20, hh - 20, 9) for x in np.linspace(20, ww - 20, 11)], np.float32)
    guess = pts + np.array([-5.0 / scale, 3.0 / scale], np.float32)
    on, os_, _ = klt_track(oa, ob, pts, guess.copy())
    gn, gs = t.track_points(pts, guess.copy())
    assert np.array_equal(gs, os_) and np.array_equal(gn, on)
    assert gs.sum() > 60
    # and replenishment with landmarks present (occupancy circles around them)
    more = v.replenishFeatures()
    st = v.tc_ekf.get_state()
    Ks = Kc.copy()
    if scale > 1:
        for i in (0, 2, 4, 5):
            Ks[i] = np.float32(np.float64(Kc[i]) / scale)
    ex = np.stack([st["feat_mu"][:len(px), 0] * Ks[0], st["feat_mu"][:len(px), 1] * Ks[4]], axis=1).astype(np.float32)
    ref2 = replenish(rb, ex, 200)
    assert np.array_equal(more, ref2)
    v.tc_ekf.close()


def test_node_accepts_a_1280x960_sequence_without_a_config_edit():
    """the Python node on a 1280x960 sequence with the node's default scale 4, default capacity."""
    a = big_image(1280, 960)
    seq = translated_sequence(a, 4, dx=-4.0, dy=-2.0)
    Kc = np.array([900.0, 0, 640.0, 0, 900.0, 480.0, 0, 0, 1.0], np.float32)
    v = EKFVIO(max_features=100, inverse_image_scale=4, replenish=1)
    node = OracleNode(100, Kc, inverse_image_scale=4)
    for i, img in enumerate(seq):
        if i > 0:
            v.tc_ekf.set_state(node.ekf.get_state())
        assert v.addFrame(3.0 + i / 30.0, img, Kc) in (capi.OK, capi.ENUMERIC)
        node.add_frame(3.0 + i / 30.0, img)
        sg, so = v.tc_ekf.get_state(), node.ekf.get_state()
        assert v.tc_ekf.num_features == node.ekf.num_features > 20
        assert np.array_equal(sg["del_flag"], so["del_flag"]) and np.array_equal(sg["last_klt"], so["last_klt"]), i
    v.tc_ekf.close()


# ------------------------------------------------------------------ (d) a device ordinal other than 0
def test_handle_on_device_one():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box")
    sc = Scenario(32, seed=1)
    out = []
    for dev in (0, 1):
        g = TightlyCoupledEKF(max_features=32, device=dev)
        g.addNewFeatures(sc.initial_features())
        s2 = Scenario(32, seed=1)
        for z, R, p in s2.frames(3):
            g.process(s2.dt)
            g.updateWithFeaturePositions(z, R, p)
        out.append(g.get_state())
        g.close()
    for k in KEYS:
        assert np.array_equal(out[0][k], out[1][k]), k
