"""ctypes driver for oracle/build/libekf_oracle.so — TEST INFRASTRUCTURE ONLY.

Mirrors the public surface of the reference's TightlyCoupledEKF
(include/ekf_vio/TightlyCoupledEKF.h:27-68) over the C++ restatement in ekf_oracle.hpp.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "libekf_oracle.so")
_lib = None

BASE = 22


def build_oracle(force=False):
    """Compile the oracle with g++ (make -C oracle)."""
    srcs = [os.path.join(_HERE, f) for f in ("ekf_oracle.hpp", "amd_order.hpp", "ekf_oracle_capi.cpp", "klt_oracle.cpp", "fast_oracle.cpp", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def oracle_lib():
    global _lib
    if _lib is None:
        build_oracle()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(lib):
    vp, i32, u8p, ip = C.c_void_p, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_int)
    for pre, ct in (("orc32", C.c_float), ("orc64", C.c_double)):
        tp = C.POINTER(ct)
        g = lambda name: getattr(lib, pre + "_" + name)
        g("create").restype = vp
        g("create").argtypes = [C.c_double, C.c_double, C.c_double, i32]
        g("destroy").argtypes = [vp]
        g("set_variant").argtypes = [vp, i32, i32, i32]
        g("num_features").argtypes = [vp]
        g("dim").argtypes = [vp]
        g("add_features").argtypes = [vp, tp, i32]
        g("process").argtypes = [vp, ct]
        g("update").argtypes = [vp, tp, tp, u8p]
        g("update").restype = i32
        g("linearize").argtypes = [vp, ct, tp]
        g("convolve_base").argtypes = [vp, tp, ct, tp]
        g("convolve_feature").argtypes = [vp, tp, tp, ct, tp]
        g("process_noise_diag").argtypes = [vp, ct, tp]
        g("measurement_map").argtypes = [vp, u8p, ip]
        g("measurement_map").restype = i32
        g("get_state").argtypes = [vp, tp, tp, tp, u8p, tp]
        g("set_state").argtypes = [vp, i32, tp, tp, tp, u8p, tp]
        g("check_sigma").argtypes = [vp, tp, tp]
        g("imu_update").argtypes = [vp, tp, tp, ct, ct, tp]
        g("rt_gravity").argtypes = [tp, tp, tp, tp]
        g("time_steps").argtypes = [vp, i32, ct, tp, tp, u8p]
        g("time_steps").restype = C.c_double
    i16p, fpp = C.POINTER(C.c_int16), C.POINTER(C.c_float)
    lib.orc_klt_frame_create.restype = vp
    lib.orc_klt_frame_create.argtypes = [u8p, i32, i32, i32, i32, i32]
    lib.orc_klt_frame_destroy.argtypes = [vp]
    lib.orc_klt_levels.argtypes = [vp]
    lib.orc_klt_level_size.argtypes = [vp, i32, ip, ip]
    lib.orc_klt_get_level.argtypes = [vp, i32, u8p, i16p]
    lib.orc_klt_track.argtypes = [vp, vp, fpp, fpp, i32, i32, i32, C.c_float, C.c_float, i32, u8p, ip]
    lib.orc_klt_uncertainty.argtypes = [vp, vp, fpp, fpp, i32, fpp]
    lib.orc_set_threads.argtypes = [i32]
    lib.orc_max_threads.restype = i32
    lib.orc_frame_resize.argtypes = [u8p, i32, i32, i32, i32, u8p]
    lib.orc_gaussian_blur5.argtypes = [u8p, i32, i32, i32, C.c_float, u8p]
    lib.orc_gauss5_kernel.argtypes = [C.c_float, ip]
    lib.orc_fast_detect.argtypes = [u8p, i32, i32, i32, i32, i32, i32, ip, ip]
    lib.orc_circle_fill.argtypes = [u8p, i32, i32, i32, i32, i32]
    lib.orc_replenish.argtypes = [u8p, i32, i32, i32, fpp, i32, i32, i32, i32, i32, i32, ip]


def set_threads(t):
    oracle_lib().orc_set_threads(int(t))


def max_threads():
    return int(oracle_lib().orc_max_threads())


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


class OracleFilter:
    """CPU restatement of TightlyCoupledEKF.  dtype=np.float32 is the reference
    precision; np.float64 is the yardstick."""

    def __init__(self, dtype=np.float32, depth=0.5, depth_var=100.0, homog_var=1e-5, emulate_static_cache=True,
                 eigen_sse_quat=None, trig_float=None, div_reciprocal=None, ldlt_amd_order=None, amd_keep_diagonal=None):
        self.lib = oracle_lib()
        self.dtype = np.dtype(dtype)
        self.pre = "orc32" if self.dtype == np.float32 else "orc64"
        self.ct = C.c_float if self.dtype == np.float32 else C.c_double
        self.h = C.c_void_p(self._f("create")(depth, depth_var, homog_var, int(bool(emulate_static_cache))))
        self.set_variant(eigen_sse_quat, trig_float, div_reciprocal)
        self.set_ldlt_order(ldlt_amd_order, amd_keep_diagonal)

    def _f(self, name):
        return getattr(self.lib, self.pre + "_" + name)

    def set_variant(self, eigen_sse_quat=None, trig_float=None, div_reciprocal=None):
        """Which x86-64 Eigen build the restatement follows (ekf_oracle.hpp Config): SSE2 quaternion product and
        reduction order (default on), float sin/cos (default off), `/=` as a reciprocal multiply (default off).
        None leaves a switch unchanged."""
        enc = lambda v: -1 if v is None else int(bool(v))
        self._f("set_variant")(self.h, enc(eigen_sse_quat), enc(trig_float), enc(div_reciprocal))

    def set_ldlt_order(self, ldlt_amd_order=None, amd_keep_diagonal=None, general_path=None):
        """SimplicialLDLT's ordering (ekf_oracle.hpp Config): Eigen's default AMD ordering on S's structural pattern (default on;
        off = natural order, rounds 1-5), with the diagonal in the pattern as Eigen hands it over (default on).  None: unchanged."""
        enc = lambda v: -1 if v is None else int(bool(v))
        fn = self._f("set_ldlt_order")
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        fn.restype = None
        fn(self.h, enc(ldlt_amd_order), enc(amd_keep_diagonal), enc(general_path))

    def last_perm(self):
        """The ordering the last update's LDLT used: P[k] = measurement row of the k-th pivot."""
        fn = self._f("last_perm")
        fn.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
        fn.restype = C.c_int
        m = fn(self.h, None, 0)
        out = np.zeros(max(m, 1), np.int32)
        fn(self.h, out.ctypes.data_as(C.POINTER(C.c_int)), m)
        return out[:m]

    def close(self):
        if self.h:
            self._f("destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _arr(self, a):
        return np.ascontiguousarray(a, dtype=self.dtype)

    @property
    def num_features(self):
        return int(self._f("num_features")(self.h))

    @property
    def dim(self):
        return int(self._f("dim")(self.h))

    def add_new_features(self, uv):
        uv = self._arr(uv).reshape(-1, 2)
        self._f("add_features")(self.h, _p(uv, self.ct), uv.shape[0])

    def process(self, dt):
        self._f("process")(self.h, self.ct(dt))

    def update(self, z, R, passed):
        N = self.num_features
        z = self._arr(z).reshape(N, 2)
        R = self._arr(R).reshape(N, 4)
        p = np.ascontiguousarray(passed, dtype=np.uint8).reshape(N)
        return int(self._f("update")(self.h, _p(z, self.ct), _p(R, self.ct), _p(p, C.c_uint8)))

    def linearize(self, dt):
        n = self.dim
        F = np.zeros((n, n), dtype=self.dtype)  # filled column-major
        self._f("linearize")(self.h, self.ct(dt), _p(F, self.ct))
        return F.T.copy()  # -> F[row, col]

    def convolve_base_state(self, mu, dt):
        mu = self._arr(mu)
        out = np.zeros(BASE, dtype=self.dtype)
        self._f("convolve_base")(self.h, _p(mu, self.ct), self.ct(dt), _p(out, self.ct))
        return out

    def convolve_feature(self, base, feat, dt):
        base, feat = self._arr(base), self._arr(feat)
        out = np.zeros(3, dtype=self.dtype)
        self._f("convolve_feature")(self.h, _p(base, self.ct), _p(feat, self.ct), self.ct(dt), _p(out, self.ct))
        return out

    def process_noise_diag(self, dt):
        q = np.zeros(self.dim, dtype=self.dtype)
        self._f("process_noise_diag")(self.h, self.ct(dt), _p(q, self.ct))
        return q

    def form_feature_measurement_map(self, measured):
        m = np.ascontiguousarray(measured, dtype=np.uint8)
        idx = np.zeros(2 * len(m) + 1, dtype=np.int32)
        k = self._f("measurement_map")(self.h, _p(m, C.c_uint8), _p(idx, C.c_int))
        return idx[:k].copy()

    def get_state(self):
        N, n = self.num_features, self.dim
        base = np.zeros(BASE, self.dtype)
        feat = np.zeros((N, 3), self.dtype)
        klt = np.zeros((N, 2), self.dtype)
        dele = np.zeros(N, np.uint8)
        sig = np.zeros((n, n), self.dtype)
        self._f("get_state")(self.h, _p(base, self.ct), _p(feat, self.ct), _p(klt, self.ct), _p(dele, C.c_uint8),
                             _p(sig, self.ct))
        return dict(base_mu=base, feat_mu=feat, last_klt=klt, del_flag=dele, Sigma=sig.T.copy())

    def set_state(self, st):
        base = self._arr(st["base_mu"])
        feat = self._arr(st["feat_mu"]).reshape(-1, 3)
        N = feat.shape[0]
        klt = self._arr(st["last_klt"]).reshape(N, 2)
        dele = np.ascontiguousarray(st["del_flag"], dtype=np.uint8).reshape(N)
        sig = np.ascontiguousarray(np.asarray(st["Sigma"], dtype=self.dtype).T)  # column-major
        self._f("set_state")(self.h, N, _p(base, self.ct), _p(feat, self.ct), _p(klt, self.ct), _p(dele, C.c_uint8),
                             _p(sig, self.ct))

    def imu_update(self, gyro, accel, gyro_var=1e-4, accel_var=1e-2, gravity=(0.0, 9.81, 0.0)):
        """the IMU measurement update specified in ekf_oracle.hpp (own design; the reference stubs it)."""
        g, a, gr = self._arr(gyro), self._arr(accel), self._arr(gravity)
        self._f("imu_update")(self.h, _p(g, self.ct), _p(a, self.ct), self.ct(gyro_var), self.ct(accel_var), _p(gr, self.ct))

    def rt_gravity(self, q, g):
        """(R(q)^T g, its 3x4 Jacobian with respect to (w,x,y,z)) as the IMU update evaluates them."""
        q, g = self._arr(q), self._arr(g)
        out, jac = np.zeros(3, self.dtype), np.zeros(12, self.dtype)
        self._f("rt_gravity")(_p(q, self.ct), _p(g, self.ct), _p(out, self.ct), _p(jac, self.ct))
        return out, jac.reshape(3, 4)

    def check_sigma(self):
        a, b = self.ct(0), self.ct(0)
        self._f("check_sigma")(self.h, C.byref(a), C.byref(b))
        return float(a.value), float(b.value)

    def time_steps(self, steps, dt, z, R, passed):
        N = self.num_features
        z = self._arr(z).reshape(N, 2)
        R = self._arr(R).reshape(N, 4)
        p = np.ascontiguousarray(passed, dtype=np.uint8).reshape(N)
        return float(self._f("time_steps")(self.h, int(steps), self.ct(dt), _p(z, self.ct), _p(R, self.ct),
                                           _p(p, C.c_uint8)))


class KltFrame:
    """Pyramid + Scharr derivatives of one 8-bit frame (restated OpenCV buildOpticalFlowPyramid)."""

    def __init__(self, img, win=21, max_level=3):
        self.lib = oracle_lib()
        img = np.ascontiguousarray(img, dtype=np.uint8)
        self.h_, self.w_ = img.shape
        self.win = win
        self.p = C.c_void_p(self.lib.orc_klt_frame_create(_p(img, C.c_uint8), self.w_, self.h_, self.w_, win, max_level))

    def __del__(self):
        try:
            if self.p:
                self.lib.orc_klt_frame_destroy(self.p)
                self.p = None
        except Exception:
            pass

    @property
    def levels(self):
        return int(self.lib.orc_klt_levels(self.p))

    def level(self, l):
        w, h = C.c_int(0), C.c_int(0)
        self.lib.orc_klt_level_size(self.p, l, C.byref(w), C.byref(h))
        img = np.zeros((h.value, w.value), np.uint8)
        der = np.zeros((h.value, w.value, 2), np.int16)
        self.lib.orc_klt_get_level(self.p, l, _p(img, C.c_uint8), _p(der, C.c_int16))
        return img, der


def klt_track(prev, nxt, prev_px, init_px, win=21, max_iter=30, epsilon=0.01, min_eig=1e-4, accum_mode=0):
    """calcOpticalFlowPyrLK(prev, next, prev_px, init_px, ..., OPTFLOW_USE_INITIAL_FLOW) restated.
    Returns (next_px[n,2] float32, status[n] uint8, iterations[n])."""
    lib = oracle_lib()
    pp = np.ascontiguousarray(prev_px, dtype=np.float32).reshape(-1, 2)
    nn = np.array(init_px, dtype=np.float32, order="C", copy=True).reshape(-1, 2)
    n = pp.shape[0]
    st = np.zeros(n, np.uint8)
    it = np.zeros(n, np.int32)
    lib.orc_klt_track(prev.p, nxt.p, _p(pp, C.c_float), _p(nn, C.c_float), n, win, max_iter, epsilon, min_eig,
                      accum_mode, _p(st, C.c_uint8), it.ctypes.data_as(C.POINTER(C.c_int)))
    return nn, st, it


def klt_uncertainty(prev, cur, ref_px, cur_px):
    """KLTTracker::estimateUncertaintySampleBased (KLTTracker.cpp:111-175) restated: pixel-space 2x2 covariance per
    point from 25 SSD-weighted samples of 5x5 sub-pixel patches.  Returns cov[n,2,2] float32."""
    lib = oracle_lib()
    rp = np.ascontiguousarray(ref_px, dtype=np.float32).reshape(-1, 2)
    cp = np.ascontiguousarray(cur_px, dtype=np.float32).reshape(-1, 2)
    n = rp.shape[0]
    cov = np.zeros((n, 4), np.float32)
    lib.orc_klt_uncertainty(prev.p, cur.p, _p(rp, C.c_float), _p(cp, C.c_float), n, _p(cov, C.c_float))
    return cov.reshape(n, 2, 2)


# ---- frame ingest and landmark replenishment (fast_oracle.cpp) ----
def frame_resize(img, inv_scale):
    """Frame::Frame image part: cv::resize(img, Size(cols / s, rows / s)), INTER_LINEAR, 8-bit."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.zeros((h // inv_scale, w // inv_scale), np.uint8)
    rc = oracle_lib().orc_frame_resize(_p(img, C.c_uint8), w, h, w, int(inv_scale), _p(out, C.c_uint8))
    assert rc == 0
    return out


def gaussian_blur5(img, sigma):
    """cv::GaussianBlur(img, Size(5,5), sigma) on an 8-bit image (OpenCV 3.x fixed-point path, reflect-101)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.zeros_like(img)
    rc = oracle_lib().orc_gaussian_blur5(_p(img, C.c_uint8), w, h, w, float(sigma), _p(out, C.c_uint8))
    assert rc == 0
    return out


def gauss5_kernel(sigma):
    """the five fixed-point taps (x256) of that blur"""
    k = np.zeros(5, np.int32)
    oracle_lib().orc_gauss5_kernel(float(sigma), k.ctypes.data_as(C.POINTER(C.c_int)))
    return k


def fast_detect(img, threshold=50, nonmax=True):
    """cv::FAST(img, kp, threshold, nonmax), TYPE_9_16: (xy[n,2] int32 in raster order, score[n] int32)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    cap = (w * h) // 4 + 1 if nonmax else w * h
    xy = np.zeros((cap, 2), np.int32)
    sc = np.zeros(cap, np.int32)
    n = oracle_lib().orc_fast_detect(_p(img, C.c_uint8), w, h, w, int(threshold), int(bool(nonmax)), cap,
                                     xy.ctypes.data_as(C.POINTER(C.c_int)), sc.ctypes.data_as(C.POINTER(C.c_int)))
    return xy[:n].copy(), sc[:n].copy()


def circle_fill(mask, cx, cy, radius):
    """cv::circle(mask, (cx, cy), radius, 255, -1) in place."""
    assert mask.dtype == np.uint8 and mask.flags["C_CONTIGUOUS"]
    h, w = mask.shape
    oracle_lib().orc_circle_fill(_p(mask, C.c_uint8), w, h, int(cx), int(cy), int(radius))
    return mask


def replenish(img, existing_px, num_features, threshold=50, min_dist=30, kill_pad=11):
    """EKFVIO::replenishFeatures on a scaled frame: pixels (int32 [k,2]) of the landmarks to add."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    ex = np.ascontiguousarray(existing_px, dtype=np.float32).reshape(-1, 2)
    cap = max(int(num_features), 1)
    out = np.zeros((cap, 2), np.int32)
    n = oracle_lib().orc_replenish(_p(img, C.c_uint8), w, h, w, _p(ex, C.c_float), ex.shape[0], int(num_features),
                                   int(threshold), int(min_dist), int(kill_pad), cap, out.ctypes.data_as(C.POINTER(C.c_int)))
    return out[:n].copy()


def amd_order(pattern, keep_diagonal=True):
    """oracle/amd_order.hpp on a boolean (n, n) pattern (symmetrised here): P with P[k] = the k-th pivot's index."""
    a = np.asarray(pattern, bool)
    a = a | a.T
    n = a.shape[0]
    cp = np.zeros(n + 1, np.int32)
    rows = []
    for j in range(n):
        r = np.nonzero(a[:, j])[0]
        rows.append(r)
        cp[j + 1] = cp[j] + len(r)
    ri = np.ascontiguousarray(np.concatenate(rows) if n and cp[n] else np.zeros(0), np.int32)
    out = np.zeros(max(n, 1), np.int32)
    fn = oracle_lib().orc_amd_order
    fn.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
    fn.restype = None
    fn(n, cp.ctypes.data_as(C.POINTER(C.c_int)), ri.ctypes.data_as(C.POINTER(C.c_int)), int(bool(keep_diagonal)), out.ctypes.data_as(C.POINTER(C.c_int)))
    return out[:n]
