"""Frame ingest (resize) and landmark replenishment (FAST-9/16 + occupancy first-fit) oracle: checks that pin
the restatement by construction (parity against OpenCV itself is unpinned)."""
import os

import numpy as np
from PIL import Image

from oracle import circle_fill, fast_detect, frame_resize, replenish

IMG = os.path.join(os.path.dirname(__file__), "golden", "images")
CIRCLE16 = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
            (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def grey(name):
    return np.asarray(Image.open(os.path.join(IMG, name + "_gray.png")))


def test_resize_integer_scales_are_box_means():
    """cols % s == 0: the bilinear sample point of scale 2 / 4 is the centre of a 2x2 block: rounded mean."""
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (96, 128), dtype=np.uint8)
    assert np.array_equal(frame_resize(img, 1), img)
    for s in (2, 4):
        o = s // 2 - 1
        blk = img.astype(np.int32)
        ref = (blk[o::s, o::s] + blk[o::s, o + 1::s] + blk[o + 1::s, o::s] + blk[o + 1::s, o + 1::s] + 2) >> 2
        assert np.array_equal(frame_resize(img, s), ref.astype(np.uint8)), s
    # odd scale: the sample point is a pixel centre
    assert np.array_equal(frame_resize(img[:96, :126], 3), img[1:96:3, 1:126:3])
    # non-divisible size: shape is floor, values stay within the local 3x3 range
    r = frame_resize(img[:95, :127], 4)
    assert r.shape == (23, 31)


def _brute_corner(img, x, y, t):
    v = int(img[y, x])
    ring = [int(img[y + dy, x + dx]) for dx, dy in CIRCLE16]
    best = None
    for start in range(16):
        seg = [ring[(start + k) % 16] for k in range(9)]
        if all(q > v + t for q in seg) or all(q < v - t for q in seg):
            best = True
    return bool(best)


def _brute_score(img, x, y, t0):
    t = t0
    while t < 255 and _brute_corner(img, x, y, t + 1):
        t += 1
    return t


def test_fast_segment_test_and_score_against_brute_force():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (40, 48), dtype=np.uint8)
    img[10:30, 12:36] = np.where(rng.random((20, 24)) < 0.5, 20, 230).astype(np.uint8)  # strong contrast patch
    t = 40
    xy, sc = fast_detect(img, t, nonmax=False)
    found = {(int(x), int(y)): int(s) for (x, y), s in zip(xy, sc)}
    for y in range(3, 37):
        for x in range(3, 45):
            b = _brute_corner(img, x, y, t)
            assert b == ((x, y) in found), (x, y)
            if b:
                assert found[(x, y)] == _brute_score(img, x, y, t), (x, y)
    # raster order, border excluded
    assert all((xy[i][1], xy[i][0]) < (xy[i + 1][1], xy[i + 1][0]) for i in range(len(xy) - 1))
    assert xy[:, 0].min() >= 3 and xy[:, 0].max() <= 44 and xy[:, 1].min() >= 3 and xy[:, 1].max() <= 36
    # non-maximum suppression: strict maximum of the 8-neighbourhood of the score map
    smap = -np.ones(img.shape, np.int32)
    smap[xy[:, 1], xy[:, 0]] = sc
    xy2, sc2 = fast_detect(img, t, nonmax=True)
    keep = set()
    for (x, y), s in zip(xy, sc):
        nb = smap[y - 1:y + 2, x - 1:x + 2].copy()
        nb[1, 1] = -2
        if s > nb.max():
            keep.add((int(x), int(y)))
    assert keep == {(int(x), int(y)) for x, y in xy2}


def test_circle_matches_midpoint_algorithm_properties():
    m = np.zeros((101, 101), np.uint8)
    circle_fill(m, 50, 50, 30)
    assert m[50, 20] == 255 and m[50, 80] == 255 and m[20, 50] == 255 and m[80, 50] == 255
    assert m[50, 19] == 0 and m[19, 50] == 0
    assert np.array_equal(m, m.T) and np.array_equal(m, m[::-1]) and np.array_equal(m, m[:, ::-1])
    yy, xx = np.mgrid[:101, :101]
    d2 = (yy - 50) ** 2 + (xx - 50) ** 2
    assert np.all(m[d2 <= 29 ** 2] == 255) and np.all(m[d2 > 31 ** 2] == 0)
    # clipping at the border
    c = np.zeros((40, 50), np.uint8)
    circle_fill(c, 2, 37, 30)
    assert c[39, 0] == 255 and c[0, 49] == 0 and c.sum() > 0


def test_replenish_first_fit_invariants():
    img = grey("640_480_test")
    xy, sc = fast_detect(img, 50, True)
    assert len(xy) > 50
    new = replenish(img, np.zeros((0, 2), np.float32), 100)
    assert 0 < len(new) <= 100
    # accepted points are keypoints, in raster order, inside the kill box, pairwise farther apart than the circle
    kp = {(int(x), int(y)) for x, y in xy}
    assert all((int(x), int(y)) in kp for x, y in new)
    assert all((new[i][1], new[i][0]) < (new[i + 1][1], new[i + 1][0]) for i in range(len(new) - 1))
    assert np.all(new[:, 0] >= 11) and np.all(new[:, 1] >= 11) and np.all(640 - new[:, 0] >= 11) and np.all(480 - new[:, 1] >= 11)
    mask = np.zeros(img.shape, np.uint8)
    for x, y in new:
        assert mask[y, x] == 0
        circle_fill(mask, x, y, 30)
    # first fit: the first keypoint that is free and in the box is always taken
    first = next((int(x), int(y)) for x, y in xy if 11 <= x and 11 <= y and 640 - x >= 11 and 480 - y >= 11)
    assert (int(new[0][0]), int(new[0][1])) == first
    # existing landmarks block their neighbourhood; a full filter adds nothing
    ex = new[:3].astype(np.float32) + 0.4
    again = replenish(img, ex, 100)
    assert len(again) <= 97
    for x, y in again:
        assert all((x - ex_x) ** 2 + (y - ex_y) ** 2 > 29 ** 2 for ex_x, ex_y in np.rint(ex))
    assert len(replenish(img, np.zeros((100, 2), np.float32), 100)) == 0


def test_gaussian_blur5_known_answers():
    """cv::GaussianBlur(5x5, sigma) restated for 8-bit images (OpenCV 3.x fixed-point path).  Known answers from the
    published definition: taps = round(256 * normalised Gaussian), symmetric; the blur equals the float correlation
    with taps/256 under mirror (reflect-101) borders up to the final rounding; an impulse reproduces the outer product
    of the taps."""
    from oracle import gauss5_kernel, gaussian_blur5
    from scipy.ndimage import correlate1d
    k = gauss5_kernel(1.0)
    assert list(k) == [14, 63, 103, 63, 14]  # exp(-x^2/2) / sum * 256, rounded
    assert list(gauss5_kernel(0.5)) == [0, 27, 201, 27, 0]
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (37, 53)).astype(np.uint8)
    for sigma in (0.7, 1.0, 2.5):
        kk = gauss5_kernel(sigma).astype(np.float64) / 256.0
        ref = correlate1d(correlate1d(img.astype(np.float64), kk, axis=0, mode="mirror"), kk, axis=1, mode="mirror")
        got = gaussian_blur5(img, sigma).astype(np.float64)
        assert np.abs(got - np.minimum(ref, 255.0)).max() <= 0.5 + 1e-9
    imp = np.zeros((11, 11), np.uint8)
    imp[5, 5] = 255
    out = gaussian_blur5(imp, 1.0).astype(np.int64)
    want = (255 * np.outer(k, k) + (1 << 15)) >> 16
    assert np.array_equal(out[3:8, 3:8], want) and out.sum() == want.sum()
