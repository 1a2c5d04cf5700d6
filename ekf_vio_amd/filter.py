"""Host-side mirror of the reference's `class TightlyCoupledEKF`
(include/ekf_vio/TightlyCoupledEKF.h:25-68) over the C-ABI of libekfvio_hip.so.

Same member names and argument meaning as the reference so that tests read like the
reference's own (test/test_ekf.cpp, test/jacobian_test.cpp, test/analyzeEKFSimulation.cpp).
All arithmetic happens in HIP kernels on the MI355X; this file only marshals buffers.
"""
import ctypes as C

import numpy as np

from . import capi

BASE_STATE_SIZE = 22


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


class TightlyCoupledEKF:
    def __init__(self, max_features=100, device=0, stream=None, predict_mode=capi.PREDICT_STRUCTURED,
                 default_point_depth=0.5, default_point_depth_variance=100.0,
                 default_point_homogenous_variance=1e-5, hooks=False, **cfg_overrides):
        # hooks=True: this handle lives in libekfvio_hip_hooks.so, the build that also has include/ekfvio_test_hooks.h (tests, profiling scripts)
        self.hooks = bool(hooks)
        self.lib = capi.load(hooks=self.hooks)
        cfg = capi.Config()
        self._chk(self.lib.ekfvio_default_config(C.byref(cfg)))
        cfg.max_features = max_features
        cfg.predict_mode = predict_mode
        cfg.default_point_depth = default_point_depth
        cfg.default_point_depth_variance = default_point_depth_variance
        cfg.default_point_homogenous_variance = default_point_homogenous_variance
        for k, v in cfg_overrides.items():
            if k == "gravity":
                cfg.gravity[0], cfg.gravity[1], cfg.gravity[2] = (float(x) for x in v)
            else:
                setattr(cfg, k, v)
        self.cfg = cfg
        self.h = C.c_void_p()
        rc = self.lib.ekfvio_create(C.byref(cfg), device, C.c_void_p(stream) if stream else None, C.byref(self.h))
        if rc != capi.OK:
            # a failed create releases whatever it had allocated and hands back no handle
            self.h = None
            raise capi.EkfvioError(rc, "ekfvio_create failed (device %d)" % device)

    def _chk(self, rc, allow=()):
        if rc != capi.OK and rc not in allow:
            msg = self.lib.ekfvio_last_error(self.h).decode() if getattr(self, "h", None) else ""
            raise capi.EkfvioError(rc, msg)
        return rc

    def close(self):
        if getattr(self, "h", None):
            self.lib.ekfvio_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference members -------------------------------------------------------------
    def initializeBaseState(self):
        self._chk(self.lib.ekfvio_reset(self.h))

    def addNewFeatures(self, new_homogenous_features):
        uv = np.ascontiguousarray(new_homogenous_features, dtype=np.float32).reshape(-1, 2)
        self._chk(self.lib.ekfvio_add_features(self.h, _fp(uv), uv.shape[0]))

    def process(self, dt):
        self._chk(self.lib.ekfvio_process(self.h, float(dt)))

    def numericallyLinearizeProcess(self, dt):
        n = self.dim
        F = np.zeros((n, n), dtype=np.float32)
        self._chk(self.lib.ekfvio_linearize(self.h, float(dt), _fp(F)))
        return F.T.copy()

    def updateWithFeaturePositions(self, measured_positions, estimated_covariance, passed):
        """Returns capi.OK or capi.ENUMERIC (non-positive Cholesky pivot; state still updated)."""
        N = self.num_features
        z = np.ascontiguousarray(measured_positions, dtype=np.float32).reshape(-1, 2)
        R = np.ascontiguousarray(estimated_covariance, dtype=np.float32).reshape(-1, 4)
        p = np.ascontiguousarray(passed, dtype=np.uint8).reshape(-1)
        if not (z.shape[0] == R.shape[0] == p.shape[0]):
            raise capi.EkfvioError(capi.EINVAL, "size mismatch")  # ROS_ASSERT :478
        return self._chk(self.lib.ekfvio_update(self.h, _fp(z), _fp(R), _u8(p), p.shape[0]), allow=(capi.ENUMERIC,))

    def formFeatureMeasurementMap(self, measured):
        m = np.ascontiguousarray(measured, dtype=np.uint8)
        idx = np.zeros(2 * len(m) + 1, dtype=np.int32)
        rows = C.c_int32(0)
        self._chk(self.lib.ekfvio_measurement_map(self.h, _u8(m), len(m), idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                                  C.byref(rows)))
        return idx[:rows.value].copy()

    def previousFeaturePositionVector(self):
        return self.get_state()["last_klt"]

    def getFeatureHomogenousCovariance(self, index):
        cov = np.zeros(4, np.float32)
        self._chk(self.lib.ekfvio_get_feature_cov(self.h, index, _fp(cov)))
        return cov.reshape(2, 2).T.copy()

    def setFeatureHomogenousCovariance(self, index, cov):
        """TightlyCoupledEKF.cpp:668-676; cov is [row, col]."""
        c = np.ascontiguousarray(np.asarray(cov, np.float32).reshape(2, 2).T)  # column-major buffer
        self._chk(self.lib.ekfvio_set_feature_cov(self.h, int(index), _fp(c)))

    def getMetric2PixelMap(self, K):
        """TightlyCoupledEKF.cpp:683-689: diag(K(0,0), K(1,1)) as a dense 2x2."""
        K = np.ascontiguousarray(K, dtype=np.float32).reshape(9)
        J = np.zeros(4, np.float32)
        self._chk(self.lib.ekfvio_metric2pixel_map(_fp(K), _fp(J)))
        return J.reshape(2, 2).T.copy()

    def getPixel2MetricMap(self, K):
        """TightlyCoupledEKF.cpp:691-697: diag(1/K(0,0), 1/K(1,1))."""
        K = np.ascontiguousarray(K, dtype=np.float32).reshape(9)
        J = np.zeros(4, np.float32)
        self._chk(self.lib.ekfvio_pixel2metric_map(_fp(K), _fp(J)))
        return J.reshape(2, 2).T.copy()

    def getFeatureDepthVariance(self, index):
        v = C.c_float(0)
        self._chk(self.lib.ekfvio_get_depth_variance(self.h, index, C.byref(v)))
        return float(v.value)

    def imuUpdate(self, gyro, accel):
        """the IMU measurement update alone (no propagation); asynchronous."""
        g = np.ascontiguousarray(gyro, dtype=np.float32)
        a = np.ascontiguousarray(accel, dtype=np.float32)
        self._chk(self.lib.ekfvio_imu_update(self.h, _fp(g), _fp(a)))

    def checkSigma(self):
        a, b = C.c_float(0), C.c_float(0)
        self._chk(self.lib.ekfvio_check_sigma(self.h, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    # ---- state access ------------------------------------------------------------------
    @property
    def num_features(self):
        return int(self.lib.ekfvio_num_features(self.h))

    @property
    def dim(self):
        return int(self.lib.ekfvio_dim(self.h))

    @property
    def base_mu(self):
        b = np.zeros(BASE_STATE_SIZE, np.float32)
        self._chk(self.lib.ekfvio_get_base_mu(self.h, _fp(b)))
        return b

    @property
    def Sigma(self):
        n = self.dim
        s = np.zeros((n, n), np.float32)
        self._chk(self.lib.ekfvio_get_sigma(self.h, _fp(s), n))
        return s.T.copy()

    def get_state(self):
        N = self.num_features
        feat = np.zeros((N, 3), np.float32)
        klt = np.zeros((N, 2), np.float32)
        dele = np.zeros(N, np.uint8)
        self._chk(self.lib.ekfvio_get_features(self.h, _fp(feat), _fp(klt), _u8(dele)))
        return dict(base_mu=self.base_mu, feat_mu=feat, last_klt=klt, del_flag=dele, Sigma=self.Sigma)

    def set_state(self, st):
        base = np.ascontiguousarray(st["base_mu"], dtype=np.float32)
        feat = np.ascontiguousarray(st["feat_mu"], dtype=np.float32).reshape(-1, 3)
        N = feat.shape[0]
        klt = np.ascontiguousarray(st["last_klt"], dtype=np.float32).reshape(N, 2)
        dele = np.ascontiguousarray(st["del_flag"], dtype=np.uint8).reshape(N)
        sig = np.ascontiguousarray(np.asarray(st["Sigma"], dtype=np.float32).T)
        n = BASE_STATE_SIZE + 3 * N
        self._chk(self.lib.ekfvio_set_state(self.h, N, _fp(base), _fp(feat), _fp(klt), _u8(dele), _fp(sig), n))

    # ---- device-resident sequences -----------------------------------------------------
    def upload_measurements(self, z, R, passed):
        z = np.ascontiguousarray(z, dtype=np.float32)
        R = np.ascontiguousarray(R, dtype=np.float32)
        p = np.ascontiguousarray(passed, dtype=np.uint8)
        frames = p.shape[0]
        self._chk(self.lib.ekfvio_upload_measurements(self.h, frames, _fp(z), _fp(R), _u8(p)))

    def run_uploaded(self, first, count, dt):
        self._chk(self.lib.ekfvio_run_uploaded(self.h, first, count, float(dt)))

    def synchronize(self):
        """Waits for the handle's stream; returns capi.ENUMERIC once if an asynchronous run met a non-positive pivot."""
        return self._chk(self.lib.ekfvio_synchronize(self.h), allow=(capi.ENUMERIC,))

    # ---- instrumentation ---------------------------------------------------------------
    def profile(self, on):
        self._chk(self.lib.ekfvio_profile_enable(self.h, int(on)))
        if on:
            self._chk(self.lib.ekfvio_profile_reset(self.h))

    def profile_report(self):
        out = {}
        for c in range(self.lib.ekfvio_profile_count()):
            ms, n, fl = C.c_double(0), C.c_int64(0), C.c_double(0)
            self._chk(self.lib.ekfvio_profile_get(self.h, c, C.byref(ms), C.byref(n), C.byref(fl)))
            out[self.lib.ekfvio_profile_name(c).decode()] = dict(ms=ms.value, launches=n.value, flops=fl.value)
        return out

    def profile_update_gemms(self, reps=50):
        """(mean launch duration in us, flops per launch) of the two P-update GEMMs at the shape of the
        most recent update, replayed back to back from a hipGraph between two HIP events."""
        us, fl = C.c_double(0), C.c_double(0)
        self._chk(self.lib.ekfvio_profile_update_gemms(self.h, int(reps), C.byref(us), C.byref(fl)))
        return us.value, fl.value

    # ---- raw kernels -------------------------------------------------------------------
    def test_gemm(self, A, B, C0, alpha=1.0, beta=0.0, transB=True, variant=0):
        """C = beta*C0 + alpha * A @ (B.T if transB else B); arrays are numpy [row, col].  (hooks build)"""
        self._need_hooks()
        A = np.asarray(A, np.float32)
        B = np.asarray(B, np.float32)
        M, K = A.shape
        N = B.shape[0] if transB else B.shape[1]
        Ac, Bc = np.array(A.T, order="C", copy=True), np.array(B.T, order="C", copy=True)  # column-major buffers
        Cc = np.array(np.asarray(C0, np.float32).T, order="C", copy=True)  # never alias the caller's C0
        self._chk(self.lib.ekfvio_test_gemm(self.h, int(transB), M, N, K, alpha, _fp(Ac), M, _fp(Bc), B.shape[0], beta,
                                            _fp(Cc), M, int(variant)))
        return Cc.T.copy()

    def counters(self):
        """ekfvio_get_counters: dict(persistent, schur, recoveries, mode, early_output_frames, t2_updates)."""
        c = (C.c_int64 * 8)()
        self._chk(self.lib.ekfvio_get_counters(self.h, c))
        return dict(persistent=int(c[0]), schur=int(c[1]), recoveries=int(c[2]), mode=int(c[3]), early_output_frames=int(c[4]), t2_updates=int(c[5]))

    def _need_hooks(self):
        if not self.hooks:
            raise RuntimeError("this entry point exists only in the hooks build: construct the filter with hooks=True")

    def persistent_sweeps(self):
        """Diagnostic: sweeps of this handle that went out as the single persistent launch so far."""
        return self.counters()["persistent"]

    def sweep_counts(self):
        """Diagnostic: dict(persistent, schur, recoveries, mode) of this handle's Cholesky sweeps."""
        c = self.counters()
        return {k: c[k] for k in ("persistent", "schur", "recoveries", "mode")}

    def sweep_fault(self, spin_limit=0, stall_workgroup=-1):
        """Fault injection for the persistent sweep (ekfvio_test_sweep_fault; hooks build)."""
        self._need_hooks()
        self._chk(self.lib.ekfvio_test_sweep_fault(self.h, int(spin_limit), int(stall_workgroup)))

    def test_cholesky_solve(self, S, Crhs):
        """Returns (L, X = Crhs @ inv(S), info).  (hooks build)"""
        self._need_hooks()
        S = np.asarray(S, np.float32)
        Crhs = np.asarray(Crhs, np.float32)
        m, nr = S.shape[0], Crhs.shape[0]
        Sc, Cc = np.array(S.T, order="C", copy=True), np.array(Crhs.T, order="C", copy=True)
        L = np.zeros((m, m), np.float32)
        X = np.zeros((m, nr), np.float32)
        info = C.c_int32(0)
        self._chk(self.lib.ekfvio_test_cholesky_solve(self.h, m, nr, _fp(Sc), _fp(Cc), _fp(L), _fp(X), C.byref(info)))
        return L.T.copy(), X.T.copy(), info.value


class KLTTracker:
    """Host mirror of the reference's `class KLTTracker` (include/ekf_vio/KLTTracker.h:88-90).
    The two most recent frames live on the device inside the filter handle (the reference
    keeps them in EKFVIO::frame_buffer, depth 2)."""

    def __init__(self, ekf):
        self.ekf = ekf
        self.lib = ekf.lib

    def push_frame(self, img, K):
        """Frame(img, K, ...) -> device pyramid; the previous current frame becomes `lf`."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        K = np.ascontiguousarray(K, dtype=np.float32).reshape(9)
        self.ekf._chk(self.lib.ekfvio_klt_push_frame(self.ekf.h, _u8(img), w, h, w, _fp(K)))

    def findNewFeaturePositions(self):
        """(measured_positions[N,2], estimated_uncertainty[N,4], passed[N]) for the filter's
        landmarks, tracked from the previous into the current frame (KLTTracker.cpp:29-95)."""
        N = self.ekf.num_features
        z = np.zeros((N, 2), np.float32)
        R = np.zeros((N, 4), np.float32)
        p = np.zeros(N, np.uint8)
        self.ekf._chk(self.lib.ekfvio_klt_track(self.ekf.h, _fp(z), _fp(R), _u8(p)))
        return z, R, p

    def track_points(self, prev_px, init_px):
        """calcOpticalFlowPyrLK(prev, cur, prev_px, init_px, OPTFLOW_USE_INITIAL_FLOW) in pixels."""
        pp = np.ascontiguousarray(prev_px, dtype=np.float32).reshape(-1, 2)
        ii = np.ascontiguousarray(init_px, dtype=np.float32).reshape(-1, 2)
        out = np.zeros_like(pp)
        st = np.zeros(pp.shape[0], np.uint8)
        self.ekf._chk(self.lib.ekfvio_klt_track_points(self.ekf.h, _fp(pp), _fp(ii), pp.shape[0], _fp(out), _u8(st)))
        return out, st

    def uncertainty_points(self, ref_px, cur_px):
        """KLTTracker::estimateUncertaintySampleBased (KLTTracker.cpp:111-175) for arbitrary points between the two
        resident frames: cov[n, 2, 2] in px^2."""
        rp = np.ascontiguousarray(ref_px, dtype=np.float32).reshape(-1, 2)
        cp = np.ascontiguousarray(cur_px, dtype=np.float32).reshape(-1, 2)
        cov = np.zeros((rp.shape[0], 4), np.float32)
        self.ekf._chk(self.lib.ekfvio_klt_uncertainty_points(self.ekf.h, _fp(rp), _fp(cp), rp.shape[0], _fp(cov)))
        return cov.reshape(-1, 2, 2)

    def level(self, l):
        w, h = C.c_int32(0), C.c_int32(0)
        self.ekf._chk(self.lib.ekfvio_klt_get_level(self.ekf.h, l, C.byref(w), C.byref(h), None, None))
        img = np.zeros((h.value, w.value), np.uint8)
        der = np.zeros((h.value, w.value, 2), np.int16)
        self.ekf._chk(self.lib.ekfvio_klt_get_level(self.ekf.h, l, C.byref(w), C.byref(h), _u8(img),
                                                    der.ctypes.data_as(C.POINTER(C.c_int16))))
        return img, der

    def padded_level(self, l):
        """Test hook: level l with its border, as the tracker reads it."""
        w, h, b = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self.ekf._chk(self.lib.ekfvio_klt_get_level(self.ekf.h, l, C.byref(w), C.byref(h), None, None))
        self.ekf._chk(self.lib.ekfvio_test_klt_padded_level(self.ekf.h, l, C.byref(b), None, None))
        img = np.zeros((h.value + 2 * b.value, w.value + 2 * b.value), np.uint8)
        der = np.zeros((h.value + 2 * b.value, w.value + 2 * b.value, 2), np.int16)
        self.ekf._chk(self.lib.ekfvio_test_klt_padded_level(self.ekf.h, l, C.byref(b), _u8(img), der.ctypes.data_as(C.POINTER(C.c_int16))))
        return img, der, b.value


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class EKFVIO:
    """Host mirror of the step sequence of EKFVIO::addFrame / updateStateWithNewImage
    (include/ekf_vio/EKFVIO.cpp:139-219) without ROS: frames in, odometry + landmark cloud out.
    With replenish=1 in the configuration addFrame also runs replenishFeatures (FAST, EKFVIO.cpp:224-311)
    on the device; otherwise call replenishFeatures() (or addNewFeatures) yourself."""

    def __init__(self, **kw):
        self.tc_ekf = TightlyCoupledEKF(**kw)
        self.tracker = KLTTracker(self.tc_ekf)
        self._imu_queue = []      # (stamp, gyro, accel), kept in stamp order; only used with use_imu = 1
        self._t_filter = None     # the stamp the device state stands at
        self._last_K = None
        self.dropped_imu = 0

    def addFrame(self, stamp, img, K):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        K = np.ascontiguousarray(K, dtype=np.float32).reshape(9)
        self._drain_imu(float(stamp))  # IMU records up to this frame's stamp, in stamp order, before the frame itself
        rc = self.tc_ekf.lib.ekfvio_step_image(self.tc_ekf.h, float(stamp), _u8(img), w, h, w, _fp(K))
        # ekfvio_step_image moves the device clock as soon as process(dt) is enqueued; only EINVAL / ECAPACITY refuse the frame
        # before that.  The host's queue clock follows whenever the predict went out, whatever happened behind it, so that a
        # failed frame does not leave IMU records between the two stamps queued for a filter that is already past them.
        if rc not in (capi.EINVAL, capi.ECAPACITY):
            self._t_filter = float(stamp) if self._t_filter is None else max(self._t_filter, float(stamp))
            self._last_K = K.copy()
        return self.tc_ekf._chk(rc, allow=(capi.ENUMERIC,))

    def replenishFeatures(self):
        """EKFVIO::replenishFeatures (EKFVIO.cpp:224-311) on the current frame, on the device: FAST-9/16 with
        non-maximum suppression, occupancy circles, first fit, addNewFeatures.  Returns the new landmarks' pixels."""
        k = C.c_int32(0)
        px = np.zeros((max(self.tc_ekf.cfg.max_features, 1), 2), np.int32)
        self.tc_ekf._chk(self.tc_ekf.lib.ekfvio_replenish(self.tc_ekf.h, C.byref(k), _ip(px)))
        return px[:k.value].copy()

    def fast(self, threshold=50, nonmax=True):
        """cv::FAST on the current (resized) frame: (xy[n,2], score[n]) in raster order."""
        w, h = C.c_int32(0), C.c_int32(0)
        self.tc_ekf._chk(self.tc_ekf.lib.ekfvio_klt_get_level(self.tc_ekf.h, 0, C.byref(w), C.byref(h), None, None))
        cap = w.value * h.value
        xy, sc, n = np.zeros((cap, 2), np.int32), np.zeros(cap, np.int32), C.c_int32(0)
        self.tc_ekf._chk(self.tc_ekf.lib.ekfvio_fast_detect(self.tc_ekf.h, int(threshold), int(bool(nonmax)), cap, _ip(xy), _ip(sc),
                                                            C.byref(n)))
        k = min(n.value, cap)
        return xy[:k].copy(), sc[:k].copy()

    def imu_callback(self, stamp, gyro, accel):
        """EKFVIO::imu_callback (EKFVIO.cpp:113-115): a logging stub in the reference, and a no-op here unless the filter was
        created with use_imu=1.  Then the record is queued and applied (propagate to its stamp, IMU measurement update) in
        stamp order in front of the next frame whose stamp is not older: IMU messages stamped after an image reach a node
        before that image does.  A record older than the filter's time is dropped and counted (`dropped_imu`)."""
        g = np.ascontiguousarray(gyro, dtype=np.float32)
        a = np.ascontiguousarray(accel, dtype=np.float32)
        if not self.tc_ekf.cfg.use_imu:
            self.tc_ekf._chk(self.tc_ekf.lib.ekfvio_imu(self.tc_ekf.h, float(stamp), _fp(g), _fp(a)))
            return
        stamp = float(stamp)
        if self._t_filter is not None and stamp < self._t_filter:
            self.dropped_imu += 1
            return
        i = len(self._imu_queue)
        while i > 0 and self._imu_queue[i - 1][0] > stamp:
            i -= 1
        self._imu_queue.insert(i, (stamp, g, a))
        if len(self._imu_queue) > 1000:  # the reference's subscriber queue (EKFVIO.cpp:80)
            self._imu_queue.pop(0)
            self.dropped_imu += 1

    def imu_now(self, stamp, gyro, accel):
        """ekfvio_imu directly, for a caller that delivers records in stamp order itself (EKFVIO_EINVAL for an older one)."""
        g = np.ascontiguousarray(gyro, dtype=np.float32)
        a = np.ascontiguousarray(accel, dtype=np.float32)
        self.tc_ekf._chk(self.tc_ekf.lib.ekfvio_imu(self.tc_ekf.h, float(stamp), _fp(g), _fp(a)))
        self._t_filter = float(stamp)

    def _drain_imu(self, until):
        while self._imu_queue and self._imu_queue[0][0] <= until:
            stamp, g, a = self._imu_queue.pop(0)
            if self._t_filter is not None and stamp < self._t_filter:
                self.dropped_imu += 1
                continue
            self.tc_ekf._chk(self.tc_ekf.lib.ekfvio_imu(self.tc_ekf.h, stamp, _fp(g), _fp(a)))
            self._t_filter = stamp

    def insight(self):
        """publishInsight's image (EKFVIO.cpp:379-442): the resized frame as BGR8 [h, w, 3] with a green 22-pixel square
        (cv::drawMarker MARKER_SQUARE) at the pixel of every landmark that is not flagged for deletion."""
        if self._last_K is None:
            raise capi.EkfvioError(capi.ESTATE, "insight() before the first frame")
        grey, _ = self.tracker.level(0)
        h, w = grey.shape
        out = np.repeat(grey[:, :, None], 3, axis=2).copy()
        s = max(int(self.tc_ekf.cfg.inverse_image_scale), 1)
        fx = np.float32(np.float64(self._last_K[0]) / s)
        fy = np.float32(np.float64(self._last_K[4]) / s)
        st = self.tc_ekf.get_state()
        for (u, v, _), flag in zip(st["feat_mu"], st["del_flag"]):
            if flag:
                continue
            px, py = int(np.rint(np.float32(u) * fx)), int(np.rint(np.float32(v) * fy))
            for d in range(-11, 12):
                for x, y in ((px + d, py - 11), (px + d, py + 11), (px - 11, py + d), (px + 11, py + d)):
                    if 0 <= x < w and 0 <= y < h:
                        out[y, x] = (0, 255, 0)
        return out

    def odometry(self):
        """What publishOdometry sends (EKFVIO.cpp:444-477): position, orientation (w,x,y,z), twist."""
        # (ekfvio_get_odometry's four slices of the base state, taken here from one 22-float read: one array and one pointer conversion
        # per frame instead of four -- the binding's own overhead was ~5 us of a 170 us frame)
        b = np.empty(22, np.float32)
        self.tc_ekf._chk(self.tc_ekf.lib.ekfvio_get_base_mu(self.tc_ekf.h, _fp(b)))
        return dict(position=b[0:3], orientation_wxyz=b[3:7], linear=b[7:10], angular=b[10:13])

    def points(self):
        """publishPoints (EKFVIO.cpp:479-518), formed on the device: (xyz[N,3], intensity[N]) = camera-frame
        (u/rho, v/rho, 1/rho) per landmark and the current frame's byte at the landmark's pixel."""
        N = self.tc_ekf.num_features
        xyz, inten = np.empty((N, 3), np.float32), np.empty(N, np.float32)
        self.tc_ekf._chk(self.tc_ekf.lib.ekfvio_get_points(self.tc_ekf.h, _fp(xyz), _fp(inten)))
        return xyz, inten
