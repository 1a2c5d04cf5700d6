"""The reference's per-frame sequence (EKFVIO::addFrame + updateStateWithNewImage + replenishFeatures,
include/ekf_vio/EKFVIO.cpp:139-311; KLTTracker::findNewFeaturePositions, KLTTracker.cpp:29-95) assembled from the
CPU oracle's parts: Frame::Frame resize -> process(dt) -> pyramidal LK seeded by the prediction -> update -> FAST
replenishment.  TEST INFRASTRUCTURE ONLY (imports oracle/)."""
import numpy as np

from oracle import KltFrame, OracleFilter, frame_resize, klt_track, replenish


class OracleNode:
    def __init__(self, max_features, K, inverse_image_scale=1, dtype=np.float32, fast_threshold=50, min_new_feature_dist=30,
                 kill_pad=11, win=21, max_level=3, do_replenish=True, depth_var=100.0):
        self.ekf = OracleFilter(dtype, depth_var=depth_var)
        self.max_features = int(max_features)
        self.scale = int(inverse_image_scale)
        K = np.asarray(K, np.float32).reshape(9).copy()
        if self.scale > 1:  # Frame.cpp:26-33: K(0,0), K(0,2), K(1,1), K(1,2) divided by the scale (in double, narrowed)
            for i in (0, 2, 4, 5):
                K[i] = np.float32(np.float64(K[i]) / self.scale)
        self.K = K
        self.thr, self.min_dist, self.kill_pad = fast_threshold, min_new_feature_dist, kill_pad
        self.win, self.max_level = win, max_level
        self.do_replenish = do_replenish
        self.t = None
        self.prev = self.cur = None  # (image, KltFrame)
        self.last = {}

    # Feature::metric2Pixel / pixel2Metric with the K indexing quirk (Feature.h:60-66): cx = cy = 0
    def _px(self, uv):
        return np.stack([uv[:, 0].astype(np.float32) * self.K[0], uv[:, 1].astype(np.float32) * self.K[4]], axis=1).astype(np.float32)

    def _metric(self, px):
        px = np.asarray(px, np.float32)
        return np.stack([px[:, 0] / self.K[0], px[:, 1] / self.K[4]], axis=1).astype(np.float32)

    def measure(self):
        """findNewFeaturePositions on (prev, cur) from the filter's present state: (z, R, pass, next_px, status)."""
        st = self.ekf.get_state()
        prev_px = self._px(st["last_klt"])
        init_px = self._px(st["feat_mu"][:, :2])
        nxt, status, _ = klt_track(self.prev[1], self.cur[1], prev_px, init_px, win=self.win)
        h, w = self.cur[0].shape
        kp = self.kill_pad
        inbox = ~((nxt[:, 0] < kp) | (nxt[:, 1] < kp) | (w - nxt[:, 0] < kp) | (h - nxt[:, 1] < kp))
        passed = ((status == 1) & inbox).astype(np.uint8)
        z = self._metric(nxt)
        N = len(passed)
        R = np.zeros((N, 4), np.float32)
        R[:, 0] = np.float32(1e-5) * np.float32((1.0 / np.float64(self.K[0])) ** 2)
        R[:, 3] = np.float32(1e-5) * np.float32((1.0 / np.float64(self.K[4])) ** 2)
        R[passed == 0] = 0
        return z, R, passed, nxt, status

    def replenish(self):
        """replenishFeatures on the current frame; returns the new landmarks' pixels (int32 [k, 2])."""
        st = self.ekf.get_state()
        ex = self._px(st["feat_mu"][:, :2]) if len(st["feat_mu"]) else np.zeros((0, 2), np.float32)
        px = replenish(self.cur[0], ex, self.max_features, threshold=self.thr, min_dist=self.min_dist, kill_pad=self.kill_pad)
        if len(px):
            self.ekf.add_new_features(self._metric(px.astype(np.float32)))
        return px

    def add_frame(self, stamp, img, measurement=None):
        """One addFrame.  Returns the update's status (0 ok, 1 numeric warning) or None when no update ran."""
        img = np.ascontiguousarray(img, np.uint8)
        frame = frame_resize(img, self.scale) if self.scale > 1 else img
        self.prev, self.cur = self.cur, (frame, KltFrame(frame, win=self.win, max_level=self.max_level))
        self.last = {}
        rc = None
        if self.prev is None:
            if self.t is None:
                self.t = stamp
        else:
            dt = np.float32(np.float64(stamp) - np.float64(self.t))
            self.ekf.process(dt)
            self.t = stamp
            if self.ekf.num_features:
                z, R, p, nxt, status = measurement if measurement is not None else self.measure()
                self.last = dict(z=z, R=R, passed=p, next_px=nxt, status=status, pre_update=self.ekf.get_state())
                rc = self.ekf.update(z, R, p)
        self.last["new_px"] = self.replenish() if self.do_replenish else np.zeros((0, 2), np.int32)
        return rc
