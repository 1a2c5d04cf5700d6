"""KLT track kernel time vs iteration budget (separates the per-level template work from the Gauss-Newton iterations)."""
import os, sys, time
import numpy as np
from PIL import Image
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import EKFVIO  # noqa: E402

IMG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "images")
a = np.asarray(Image.open(os.path.join(IMG, "640_480_test_gray.png")))
b = np.asarray(Image.open(os.path.join(IMG, "640_480_moved_test_gray.png")))
K = np.array([500.0, 0, 320.0, 0, 500.0, 240.0, 0, 0, 1.0], np.float32)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
xs, ys = np.linspace(80, 560, 16), np.linspace(60, 420, n // 16)
pts = np.array([[x, y] for y in ys for x in xs], np.float32)
for iters in (1, 2, 5, 30):
    v = EKFVIO(max_features=n, klt_max_iterations=iters)
    v.tracker.push_frame(a, K)
    v.tracker.push_frame(b, K)
    v.tc_ekf.profile(True)
    for _ in range(20):
        out, st = v.tracker.track_points(pts, pts)
    rep = v.tc_ekf.profile_report()
    print("points %d  max_iter %2d: track kernel %.1f us/call  (status ok %d)" % (n, iters, 1e3 * rep["klt_track"]["ms"] / rep["klt_track"]["launches"], int(st.sum())))
    v.tc_ekf.close()

import ctypes as C
v = EKFVIO(max_features=n)
v.tracker.push_frame(a, K)
v.tracker.push_frame(b, K)
v.tc_ekf.lib.ekfvio_test_sweep_stamps(v.tc_ekf.h, 1, None)
out, st = v.tracker.track_points(pts, pts)
buf = (C.c_int64 * 1024)()
v.tc_ekf.lib.ekfvio_test_sweep_stamps(v.tc_ekf.h, 1, buf)
s_ = list(buf)[900:940]
print("point 0 (cycles): kernel start -> first level %d" % (s_[1 + 8 * 3] - s_[0]))
for lv in (3, 2, 1, 0):
    o = 8 * lv
    print("  level %d: staging loads %6d  template + A %6d  %2d iterations %6d (%.0f each)" %
          (lv, s_[2 + o] - s_[1 + o], s_[3 + o] - s_[2 + o], s_[5 + o], s_[4 + o] - s_[3 + o], (s_[4 + o] - s_[3 + o]) / max(s_[5 + o], 1)))
