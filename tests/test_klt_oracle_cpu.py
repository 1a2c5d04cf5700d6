"""Pins the KLT oracle (oracle/klt_oracle.cpp, a restatement of OpenCV 3.x pyramidal LK as
called at KLTTracker.cpp:61-64) on the reference's own test images (images/640_480_test.png
and its translated / sheared copies, stored as 8-bit grey fixtures) and on exact integer
properties of the pyramid."""
import os

import numpy as np
import pytest
from PIL import Image

from oracle import KltFrame, klt_track

IMG = os.path.join(os.path.dirname(__file__), "golden", "images")


def grey(name):
    return np.asarray(Image.open(os.path.join(IMG, name + "_gray.png")))


def grid_points():
    xs, ys = np.linspace(80, 560, 8), np.linspace(60, 420, 8)
    return np.array([[x, y] for y in ys for x in xs], np.float32)


def test_pyramid_sizes_and_level_cutoff():
    a = KltFrame(grey("640_480_test"))
    assert a.levels == 4
    assert [a.level(l)[0].shape for l in range(4)] == [(480, 640), (240, 320), (120, 160), (60, 80)]
    # inverse_image_scale 4 (Params.h:28): 160x120 -> 80x60, 40x30; 20x15 is not larger than the 21 px window
    small = KltFrame(grey("640_480_test")[::4, ::4].copy())
    assert small.levels == 3
    odd = KltFrame(grey("640_480_test")[:251, :333].copy())
    assert [odd.level(l)[0].shape for l in range(odd.levels)] == [(251, 333), (126, 167), (63, 84), (32, 42)]


def test_pyramid_of_constant_image_is_constant_with_zero_gradient():
    img = np.full((97, 131), 77, np.uint8)
    f = KltFrame(img)
    for l in range(f.levels):
        im, de = f.level(l)
        assert np.all(im == 77) and np.all(de == 0)


def test_scharr_of_ramp():
    """I(x,y) = 2x + 3y: Scharr dx = 2*2*16 = 64, dy = 3*2*16 = 96 away from the border."""
    y, x = np.mgrid[0:40, 0:50]
    f = KltFrame((2 * x + 3 * y).astype(np.uint8), win=5, max_level=0)
    _, de = f.level(0)
    assert np.all(de[1:-1, 1:-1, 0] == 64) and np.all(de[1:-1, 1:-1, 1] == 96)


def test_reference_image_pair_translation():
    """images/640_480_moved_test.png is 640_480_test.png translated by (-21, -7) px
    (SURVEY.md section 2 row 9): the tracker, started at zero flow, must find it."""
    A, B = KltFrame(grey("640_480_test")), KltFrame(grey("640_480_moved_test"))
    pts = grid_points()
    nxt, st, it = klt_track(A, B, pts, pts.copy())
    ok = st == 1
    assert ok.sum() >= 62
    flow = (nxt - pts)[ok]
    assert np.abs(np.median(flow, axis=0) - np.array([-21.0, -7.0])).max() < 2e-3
    assert np.mean(np.abs(flow - np.array([-21.0, -7.0]))) < 0.02
    assert it.max() <= 4 * 30
    # started at the true flow it stays there
    nxt2, st2, _ = klt_track(A, B, pts, pts + np.array([-21.0, -7.0], np.float32))
    assert np.mean(np.abs((nxt2 - pts)[st2 == 1] - np.array([-21.0, -7.0]))) < 0.02


def test_identity_pair_has_zero_flow_and_bookkeeping():
    A = KltFrame(grey("640_480_test"))
    pts = grid_points()
    nxt, st, _ = klt_track(A, A, pts, pts.copy())
    assert np.abs(nxt - pts)[st == 1].max() < 1e-3
    # a point whose window lies outside the level is rejected at level 0; a flat patch fails minEig
    far = np.array([[-40.0, 100.0], [700.0, 100.0]], np.float32)
    _, st2, _ = klt_track(A, A, far, far.copy())
    assert list(st2) == [0, 0]
    flat = KltFrame(np.full((120, 160), 90, np.uint8))
    _, st3, _ = klt_track(flat, flat, [[80.0, 60.0]], [[80.0, 60.0]])
    assert list(st3) == [0]


def test_integer_and_float_accumulation_agree():
    """accum_mode 1 = OpenCV's scalar float accumulation order; mode 0 = exact int64."""
    A, B = KltFrame(grey("640_480_test")), KltFrame(grey("640_480_shear_test"))
    pts = grid_points()
    n0, s0, _ = klt_track(A, B, pts, pts.copy(), accum_mode=0)
    n1, s1, _ = klt_track(A, B, pts, pts.copy(), accum_mode=1)
    both = (s0 == 1) & (s1 == 1)
    assert (s0 != s1).sum() <= 1 and np.abs(n0 - n1)[both].max() < 0.02


def test_sample_based_uncertainty_known_answers():
    """estimateUncertaintySampleBased restated (KLTTracker.cpp:111-175).  Known answers that follow from the text of
    the function alone: on a constant image every sample matches (ssd = 0, weight exp(0) = 1), so the covariance is
    the second moment of the 5x5 sample grid {-10,-5,0,5,10}^2: diag(50, 50) exactly; a strong isolated blob makes
    the centre sample dominate (covariance -> 0); and on an image that varies along x only, the samples along y all
    match, which leaves the y variance at 50 and shrinks the x variance."""
    from oracle import klt_uncertainty
    flat = np.full((80, 100), 77, np.uint8)
    F = KltFrame(flat)
    pts = np.array([[50.0, 40.0], [20.25, 30.75], [2.0, 2.0], [99.0, 79.0]], np.float32)
    c = klt_uncertainty(F, F, pts, pts)
    assert np.array_equal(c, np.tile(np.array([[50.0, 0.0], [0.0, 50.0]], np.float32), (4, 1, 1)))
    yy, xx = np.mgrid[0:80, 0:100]
    blob = (255.0 * np.exp(-((xx - 50) ** 2 + (yy - 40) ** 2) / 8.0)).astype(np.uint8)
    Bf = KltFrame(blob)
    cb = klt_uncertainty(Bf, Bf, [[50.0, 40.0]], [[50.0, 40.0]])[0]
    assert cb[0, 0] < 1e-3 and cb[1, 1] < 1e-3 and cb[0, 1] == cb[1, 0]
    ramp = np.tile((np.arange(100) * 2.5).astype(np.uint8), (80, 1))
    Rf = KltFrame(ramp)
    cr = klt_uncertainty(Rf, Rf, [[50.5, 40.0]], [[50.5, 40.0]])[0]
    assert abs(cr[1, 1] - 50.0) < 1e-4 and cr[0, 0] < 30.0 and abs(cr[0, 1]) < 1e-4
