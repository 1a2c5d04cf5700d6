"""GPU KLT parity (BASELINE config 1: 640x480 test images, KLT + N=64-landmark update).
The HIP pyramid / Scharr / tracker are compared with the oracle bit for bit (integer images
and derivatives; tracked positions are float32 results of identical scalar arithmetic on
exactly accumulated integer sums)."""
import os

import numpy as np
import pytest
from PIL import Image

from ekf_vio_amd import EKFVIO, KLTTracker, TightlyCoupledEKF, capi, EkfvioError
from oracle import KltFrame, OracleFilter, klt_track, klt_uncertainty

from _scatter import backward_yardstick

pytestmark = pytest.mark.gpu
IMG = os.path.join(os.path.dirname(__file__), "golden", "images")
K = np.array([500.0, 0, 320.0, 0, 500.0, 240.0, 0, 0, 1.0], np.float32)  # fx=fy=500


def grey(name):
    return np.asarray(Image.open(os.path.join(IMG, name + "_gray.png")))


def grid_points(n=8):
    xs, ys = np.linspace(80, 560, n), np.linspace(60, 420, n)
    return np.array([[x, y] for y in ys for x in xs], np.float32)


@pytest.mark.parametrize("crop", [None, (251, 333), (120, 160)])
def test_pyramid_and_derivatives_bit_exact(crop):
    img = grey("640_480_test")
    if crop:
        img = img[:crop[0], :crop[1]].copy()
    g = TightlyCoupledEKF(max_features=4)
    t = KLTTracker(g)
    t.push_frame(img, K)
    o = KltFrame(img)
    for l in range(o.levels):
        gi, gd = t.level(l)
        oi, od = o.level(l)
        assert np.array_equal(gi, oi), ("image", l)
        assert np.array_equal(gd, od), ("deriv", l)
    with pytest.raises(EkfvioError):
        t.level(o.levels)  # no such level on the device either
    g.close()


def test_padded_levels_hold_reflect101_borders_whatever_the_buffers_held_before():
    """What the tracker reads: every level with a 24-pixel border, the image mirrored (BORDER_REFLECT_101, what OpenCV's
    pyramid border holds), the derivatives zero.  A larger frame first, so the planes are dirty where the smaller frames'
    borders (and, with the smaller pitch, everything else) come to lie; sizes that are no multiple of the 32-pixel tile,
    and levels narrower than two borders (several reflections)."""
    full = grey("640_480_test")
    g = TightlyCoupledEKF(max_features=4, hooks=True)
    t = KLTTracker(g)
    for crop in [(480, 640), (251, 333), (97, 61), (120, 160), (23, 70), (480, 640)]:
        img = full[:crop[0], :crop[1]].copy()
        t.push_frame(img, K)
        o = KltFrame(img)
        for l in range(o.levels):
            gi, gd, b = t.padded_level(l)
            oi, od = o.level(l)
            assert b == 24
            assert np.array_equal(gi, np.pad(oi, b, mode="reflect")), (crop, l)
            want = np.zeros_like(gd)
            want[b:-b, b:-b] = od
            assert np.array_equal(gd, want), (crop, l)
    g.close()


@pytest.mark.parametrize("second", ["640_480_moved_test", "640_480_shear_test", "640_480_test"])
def test_tracked_points_bit_exact_and_flow(second):
    a, b = grey("640_480_test"), grey(second)
    g = TightlyCoupledEKF(max_features=256)
    t = KLTTracker(g)
    t.push_frame(a, K), t.push_frame(b, K)
    pts = np.vstack([grid_points(8), grid_points(13)[:150],
                     [[5.0, 5.0], [-40.0, 100.0], [700.0, 100.0], [639.0, 479.0], [320.5, 0.25]]]).astype(np.float32)
    A, B = KltFrame(a), KltFrame(b)
    on, os_, _ = klt_track(A, B, pts, pts.copy())
    gn, gs = t.track_points(pts, pts.copy())
    assert np.array_equal(gs, os_)
    assert np.array_equal(gn, on), float(np.abs(gn - on).max())
    if second == "640_480_moved_test":
        ok = gs[:64] == 1
        assert ok.sum() >= 62
        assert np.abs(np.median((gn - pts)[:64][ok], axis=0) - np.array([-21.0, -7.0])).max() < 2e-3
    g.close()


@pytest.mark.parametrize("win,levels,iters", [(3, 3, 30), (7, 3, 30), (15, 2, 8), (19, 1, 30), (21, 0, 3)])
def test_other_windows_levels_and_iteration_budgets_bit_exact(win, levels, iters):
    """The tracker's lane geometry is three lanes per window row with runs of ceil(win / 3) pixels: every odd window up to
    the reference's 21 (Params.h) must give the oracle's positions and status, with fewer pyramid levels and a smaller
    iteration budget as well."""
    a, b = grey("640_480_test"), grey("640_480_moved_test")
    g = TightlyCoupledEKF(max_features=256, klt_window_size=win, klt_max_pyramid_level=levels, klt_max_iterations=iters)
    t = KLTTracker(g)
    t.push_frame(a, K), t.push_frame(b, K)
    pts = np.vstack([grid_points(10), [[5.0, 5.0], [-40.0, 100.0], [639.0, 479.0], [320.5, 0.25], [2.0, 470.0]]]).astype(np.float32)
    guess = (pts + np.array([-20.0, -6.0], np.float32)).astype(np.float32)  # a coarse prediction, as the filter would give
    A, B = KltFrame(a, win=win, max_level=levels), KltFrame(b, win=win, max_level=levels)
    on, os_, _ = klt_track(A, B, pts, guess.copy(), win=win, max_iter=iters)
    gn, gs = t.track_points(pts, guess.copy())
    assert np.array_equal(gs, os_)
    assert np.array_equal(gn, on), float(np.abs(gn - on).max())
    assert gs.sum() >= 60
    g.close()


def test_small_image_reduces_levels_and_flat_patch_fails():
    img = grey("640_480_test")[::4, ::4].copy()  # 160x120: level 3 would be 20x15 <= window
    g = TightlyCoupledEKF(max_features=8)
    t = KLTTracker(g)
    t.push_frame(img, K), t.push_frame(np.roll(img, 2, axis=1), K)
    o1, o2 = KltFrame(img), KltFrame(np.roll(img, 2, axis=1))
    assert o1.levels == 3
    pts = np.array([[40.0, 40.0], [80.0, 60.0], [120.0, 90.0]], np.float32)
    on, os_, _ = klt_track(o1, o2, pts, pts.copy())
    gn, gs = t.track_points(pts, pts.copy())
    assert np.array_equal(gs, os_) and np.array_equal(gn, on)
    flat = np.full((120, 160), 90, np.uint8)
    t.push_frame(flat, K), t.push_frame(flat, K)
    _, st = t.track_points([[80.0, 60.0]], [[80.0, 60.0]])
    assert list(st) == [0]
    g.close()


def test_klt_requires_two_frames():
    g = TightlyCoupledEKF(max_features=4)
    t = KLTTracker(g)
    g.addNewFeatures([[0.1, 0.1]])
    with pytest.raises(EkfvioError) as e:
        t.findNewFeaturePositions()
    assert e.value.code == capi.ESTATE
    g.close()


def _metric(px):
    """Feature::pixel2Metric with the reference's K indexing quirk (cx = cy = 0), Feature.h:60-62."""
    return np.stack([px[:, 0] / K[0], px[:, 1] / K[4]], axis=1).astype(np.float32)


def test_config1_klt_plus_64_landmark_update():
    """addFrame sequence (EKFVIO.cpp:139-219) on the reference image pair with 64 landmarks:
    first frame stores image + stamp, second frame runs process(dt) -> KLT -> update."""
    a, b = grey("640_480_test"), grey("640_480_moved_test")
    px = grid_points(8)
    uv = _metric(px)
    v = EKFVIO(max_features=64)
    assert v.addFrame(10.0, a, K) == capi.OK
    v.tc_ekf.addNewFeatures(uv)  # stands in for replenishFeatures
    v.imu_callback(10.01, [0, 0, 0], [0, 0, 9.8])  # logging stub in the reference
    o = OracleFilter(np.float32)
    o.add_new_features(uv)
    rc = v.addFrame(10.0 + 1.0 / 30.0, b, K)
    assert rc in (capi.OK, capi.ENUMERIC)
    # oracle sequence: process, KLT (prev = last_klt in pixels, init = predicted landmark), update
    dt = np.float32(np.float64(10.0 + 1.0 / 30.0) - np.float64(10.0))
    o.process(dt)
    st = o.get_state()
    prev_px = np.stack([st["last_klt"][:, 0] * K[0], st["last_klt"][:, 1] * K[4]], axis=1).astype(np.float32)
    init_px = np.stack([K[0] * st["feat_mu"][:, 0], K[4] * st["feat_mu"][:, 1]], axis=1).astype(np.float32)
    nxt, status, _ = klt_track(KltFrame(a), KltFrame(b), prev_px, init_px)
    kp = 11
    passed = (status == 1) & ~((nxt[:, 0] < kp) | (nxt[:, 1] < kp) | (640 - nxt[:, 0] < kp) | (480 - nxt[:, 1] < kp))
    z = _metric(nxt)
    R = np.zeros((64, 4), np.float32)
    R[:, 0] = np.float32(1e-5) * np.float32((1.0 / np.float64(K[0])) ** 2)
    R[:, 3] = np.float32(1e-5) * np.float32((1.0 / np.float64(K[4])) ** 2)
    o64 = OracleFilter(np.float64)
    o64.set_state(st)
    o.update(z, R, passed.astype(np.uint8)), o64.update(z, R, passed.astype(np.uint8))
    sg, so, s64 = v.tc_ekf.get_state(), o.get_state(), o64.get_state()
    assert passed.sum() >= 60
    assert np.array_equal(sg["del_flag"], so["del_flag"])           # same landmarks failed
    assert np.array_equal(sg["last_klt"], so["last_klt"])           # KLT results bit-exact through the ABI
    eg = np.abs(sg["base_mu"].astype(np.float64) - s64["base_mu"]).max()
    eo = np.abs(so["base_mu"].astype(np.float64) - s64["base_mu"]).max()
    # first update from the raw prior (cond(S) = 1.6e6): the HIP result must lie within what an
    # 8-ulp componentwise backward error on S and Sigma H^T explains (tests/_scatter.py; one ulp
    # already moves the exact base state by 2e-4 here, the fp32 oracle lands at 5e-5..9e-5)
    yard = backward_yardstick(st, z, R, passed.astype(np.uint8), s64, c=8.0)
    assert eo <= yard["mu"]
    assert eg <= yard["mu"] + 2e-5, (eg, eo, yard)
    rel = lambda x, y: np.linalg.norm(x.astype(np.float64) - y) / np.linalg.norm(y)
    assert rel(sg["Sigma"], s64["Sigma"]) <= yard["sig"] + 4 * rel(so["Sigma"], s64["Sigma"]) + 2e-6, yard
    od = v.odometry()
    assert od["position"].shape == (3,) and abs(np.linalg.norm(od["orientation_wxyz"]) - 1) < 1e-6
    xyz, inten = v.points()
    assert xyz.shape == (64, 3) and inten.shape == (64,)
    v.tc_ekf.close()


def test_sample_based_uncertainty_matches_oracle():
    """KLTTracker::estimateUncertaintySampleBased (dead code in the reference) as a HIP kernel.  The
    kernel repeats the oracle's scalar arithmetic (getRectSubPix recurrence, double pow, float sums in loop order);
    only exp() comes from a different libm, so the covariances agree to a few float ulps.  Points cover the image
    interior, the replicate border, positions outside the image and the symmetric-sample case (same frame twice)."""
    a, b = grey("640_480_test"), grey("640_480_moved_test")
    g = TightlyCoupledEKF(max_features=256)
    t = KLTTracker(g)
    t.push_frame(a, K), t.push_frame(b, K)
    ref = np.vstack([grid_points(8), grid_points(12)[:100] + np.float32(0.37),
                     [[1.0, 1.0], [638.5, 478.5], [3.99, 240.0], [320.0, 2.0], [-30.0, 50.0], [700.0, 500.0]]]).astype(np.float32)
    cur = ref + np.array([-21.0, -7.0], np.float32)
    A, B = KltFrame(a), KltFrame(b)
    oc = klt_uncertainty(A, B, ref, cur)
    gc = t.uncertainty_points(ref, cur)
    assert np.isfinite(gc).all()
    assert np.allclose(gc, oc, rtol=2e-6, atol=1e-6), float(np.abs(gc - oc).max())
    assert np.array_equal(gc[:, 0, 1], gc[:, 1, 0])
    # a well-textured, correctly tracked point is far more certain than the +-10 px sample spread of a flat one
    assert np.median(gc[:64, 0, 0]) < 50.0 and (gc[:, 0, 0] >= 0).all() and (gc[:, 1, 1] >= 0).all()
    # same frame twice, reference == sample centre: the weights are symmetric in du and dv only if the image is;
    # what must hold exactly is the centre sample's weight exp(0) = 1 bounding the sum from below
    t.push_frame(b, K)
    same = t.uncertainty_points(ref[:64], ref[:64])
    assert np.allclose(same, klt_uncertainty(B, B, ref[:64], ref[:64]), rtol=2e-6, atol=1e-6)
    g.close()


def test_sample_based_uncertainty_feeds_the_update_behind_its_flag():
    """cfg.sample_based_uncertainty = 1: ekfvio_klt_track hands the filter R = the sample covariance through the
    reference's pixel->metric conversion (KLTTracker.cpp:79-84: row 0 by (1/fx)^2, row 1 by (1/fy)^2); default 0
    keeps estimateUncertainty's constant."""
    a, b = grey("640_480_test"), grey("640_480_moved_test")
    px = grid_points(8)
    uv = _metric(px)
    out = {}
    for flag in (0, 1):
        g = TightlyCoupledEKF(max_features=64, sample_based_uncertainty=flag)
        t = KLTTracker(g)
        t.push_frame(a, K)
        g.addNewFeatures(uv)
        t.push_frame(b, K)
        z, R, p = t.findNewFeaturePositions()
        out[flag] = (z, R.reshape(-1, 4), p)
        if flag:
            st = g.get_state()
            prev_px = np.stack([st["last_klt"][:, 0] * K[0], st["last_klt"][:, 1] * K[4]], axis=1).astype(np.float32)
            nxt = np.stack([z.reshape(-1, 2)[:, 0] * K[0], z.reshape(-1, 2)[:, 1] * K[4]], axis=1).astype(np.float32)
            cov = t.uncertainty_points(prev_px, nxt).reshape(-1, 4)
        g.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][2], out[1][2])
    ok = out[1][2] == 1
    assert ok.sum() >= 60
    s0 = np.float32((1.0 / np.float64(K[0])) ** 2)
    s1 = np.float32((1.0 / np.float64(K[4])) ** 2)
    want = cov * np.array([s0, s0, s1, s1], np.float32)
    # z is metric = (px - 0) / fx: going back to pixels rounds, so the comparison is to float tolerance, not bits
    assert np.allclose(out[1][1][ok], want[ok], rtol=1e-3, atol=1e-9)
    assert np.allclose(out[0][1][ok][:, 0], np.float32(1e-5) * s0) and (out[0][1][ok][:, 1] == 0).all()
    assert (out[1][1][~ok] == 0).all()


def test_points_payload_matches_the_reference_arithmetic():
    """A19: publishPoints (EKFVIO.cpp:479-518) formed on the device: xyz = (u * z, v * z, z) with z = 1.0 / rho evaluated
    in double and narrowed, intensity = image byte at cv::Point(getPixel(f)) (round half to even; K(2) = K(5) = 0 by
    the Feature.h indexing quirk), and publishOdometry's slices of base_mu."""
    a, b = grey("640_480_test"), grey("640_480_moved_test")
    px = np.vstack([grid_points(8)[:60], [[0.5, 0.5], [638.5, 478.49], [700.0, 100.0], [-3.0, 50.0]]]).astype(np.float32)
    v = EKFVIO(max_features=64)
    assert v.points()[0].shape == (0, 3)
    v.addFrame(1.0, a, K)
    v.tc_ekf.addNewFeatures(_metric(px))
    xyz0, int0 = v.points()  # before any update: the landmarks sit exactly on their pixels of frame a
    v.addFrame(1.0 + 1.0 / 30.0, b, K)
    for img, (xyz, inten) in ((a, (xyz0, int0)), (b, v.points())):
        f = v.tc_ekf.get_state()["feat_mu"] if img is b else np.concatenate([_metric(px), np.full((64, 1), 2.0, np.float32)], axis=1)
        z = (1.0 / f[:, 2].astype(np.float64)).astype(np.float32)
        want = np.stack([f[:, 0] * z, f[:, 1] * z, z], axis=1)
        assert np.array_equal(xyz, want)
        pxf = np.float32(K[0]) * f[:, 0], np.float32(K[4]) * f[:, 1]
        ix, iy = np.rint(pxf[0]).astype(np.int64), np.rint(pxf[1]).astype(np.int64)
        inside = (ix >= 0) & (ix < 640) & (iy >= 0) & (iy < 480)
        wi = np.where(inside, img[np.clip(iy, 0, 479), np.clip(ix, 0, 639)], 0).astype(np.float32)
        assert np.array_equal(inten, wi)
        assert inside.sum() >= 60 and (~inside).sum() >= 2
    od, bm = v.odometry(), v.tc_ekf.base_mu
    assert np.array_equal(od["position"], bm[0:3]) and np.array_equal(od["orientation_wxyz"], bm[3:7])
    assert np.array_equal(od["linear"], bm[7:10]) and np.array_equal(od["angular"], bm[10:13])
    v.tc_ekf.close()


def test_step_image_with_device_side_row_count_equals_the_host_sized_update():
    """ekfvio_step_image leaves the tracker's pass flags on the device and launches the update for m = 2N rows (the true
    count is read by the kernels; the rest is identity padding).  Against the explicit sequence push_frame / process /
    findNewFeaturePositions / updateWithFeaturePositions, which sizes the launches from the host-known m, every bit of
    the state must agree -- here with 24 of 64 landmarks tracked, so the two paths really use different paddings
    (m_pad 64 vs 128)."""
    a, b = grey("640_480_test"), grey("640_480_moved_test")
    good = grid_points(8)[[9, 10, 11, 12, 13, 14, 17, 18, 19, 20, 21, 22, 25, 26, 27, 28, 29, 30, 33, 34, 35, 36, 37, 38]]
    bad = np.array([[-50.0 - 3 * i, 40.0 + 5 * i] for i in range(20)] + [[660.0 + 2 * i, 100.0 + 7 * i] for i in range(20)], np.float32)
    px = np.vstack([good[:12], bad[:20], good[12:], bad[20:]]).astype(np.float32)
    uv = _metric(px)
    t0, t1 = 3.0, 3.0 + 1.0 / 30.0
    v = EKFVIO(max_features=64)
    v.addFrame(t0, a, K)
    v.tc_ekf.addNewFeatures(uv)
    rc_a = v.addFrame(t1, b, K)
    w = TightlyCoupledEKF(max_features=64)
    tr = KLTTracker(w)
    tr.push_frame(a, K)
    w.addNewFeatures(uv)
    tr.push_frame(b, K)
    w.process(np.float32(np.float64(t1) - np.float64(t0)))
    z, R, p = tr.findNewFeaturePositions()
    assert 16 <= int(p.sum()) <= 24  # m_pad = 64 on this path
    rc_b = w.updateWithFeaturePositions(z, R, p)
    assert rc_a == rc_b
    sa, sb = v.tc_ekf.get_state(), w.get_state()
    for k in ("base_mu", "feat_mu", "last_klt", "del_flag", "Sigma"):
        assert np.array_equal(sa[k], sb[k]), k
    assert sa["del_flag"].sum() == 64 - int(p.sum())
    v.tc_ekf.close(), w.close()
