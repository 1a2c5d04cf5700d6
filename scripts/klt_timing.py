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
    v = EKFVIO(max_features=n, klt_max_iterations=iters, hooks=True)
    v.tracker.push_frame(a, K)
    v.tracker.push_frame(b, K)
    v.tc_ekf.profile(True)
    for _ in range(20):
        out, st = v.tracker.track_points(pts, pts)
    rep = v.tc_ekf.profile_report()
    print("points %d  max_iter %2d: track kernel %.1f us/call  (status ok %d)" % (n, iters, 1e3 * rep["klt_track"]["ms"] / rep["klt_track"]["launches"], int(st.sum())))
    v.tc_ekf.close()

import ctypes as C
v = EKFVIO(max_features=n, hooks=True)
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

# phase stamps of the pyramid kernel (workgroup (2,2), an interior tile)
v.tc_ekf.lib.ekfvio_test_sweep_stamps(v.tc_ekf.h, 1, None)
v.tracker.push_frame(a, K)
v.tc_ekf.lib.ekfvio_test_sweep_stamps(v.tc_ekf.h, 1, buf)
p = list(buf)[940:950]
names = ["level-0 region loads", "level-0 stores", "pyrDown 1", "level-1 stores", "level 2", "level 3"]
print("pyramid kernel, workgroup (2,2) (cycles): " + "  ".join("%s %d" % (names[i], p[i + 1] - p[i]) for i in range(6)) + "  | total %d" % (p[6] - p[0]))
import numpy as np
t = np.array(list(buf)[:600], np.int64).reshape(300, 2)
t0 = t[:, 0].min()
st_, en_ = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0  # s_memrealtime: 100 MHz
print("  all 300 workgroups (us after the first started): starts min %.2f median %.2f max %.2f; ends min %.2f median %.2f max %.2f; durations min %.2f median %.2f max %.2f"
      % (st_.min(), np.median(st_), st_.max(), en_.min(), np.median(en_), en_.max(), (en_ - st_).min(), np.median(en_ - st_), (en_ - st_).max()))
d = (en_ - st_).reshape(15, 20)
print("  durations by tile row (us): " + " ".join("%.1f" % x for x in d.mean(axis=1)))
print("  durations of tile row 0: " + " ".join("%.1f" % x for x in d[0]))

# the node's default geometry: a 640x480 upload resized by 4 inside the pyramid kernel (20 workgroups, 3 levels)
v4 = EKFVIO(max_features=n, inverse_image_scale=4, hooks=True)
v4.tracker.push_frame(a, K)
v4.tc_ekf.lib.ekfvio_test_sweep_stamps(v4.tc_ekf.h, 1, None)
v4.tracker.push_frame(b, K)
v4.tc_ekf.lib.ekfvio_test_sweep_stamps(v4.tc_ekf.h, 1, buf)
p = list(buf)[940:950]
print("pyramid kernel with the resize by 4, workgroup (2,2) (cycles): " + "  ".join("%s %d" % (names[i], p[i + 1] - p[i]) for i in range(6)) + "  | total %d" % (p[6] - p[0]))
t = np.array(list(buf)[:40], np.int64).reshape(20, 2)
t0 = t[:, 0].min()
print("  all 20 workgroups (us): ends min %.2f max %.2f; durations min %.2f median %.2f max %.2f" % (((t[:, 1] - t0) / 100.0).min(), ((t[:, 1] - t0) / 100.0).max(), ((t[:, 1] - t[:, 0]) / 100.0).min(), np.median((t[:, 1] - t[:, 0]) / 100.0), ((t[:, 1] - t[:, 0]) / 100.0).max()))
