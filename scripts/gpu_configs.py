"""Timing of the other BASELINE configurations on one MI355X: N=64 / N=1024 filter loop and
the KLT front end on the reference image pair (run on the GPU box)."""
import json
import os
import sys
import time

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import EKFVIO, KLTTracker, TightlyCoupledEKF  # noqa: E402
from ekf_vio_amd.sim import Scenario  # noqa: E402

out = {}
for N, steps in ((64, 400), (256, 400), (1024, 40)):
    sc = Scenario(N, seed=0)
    g = TightlyCoupledEKF(max_features=N)
    g.addNewFeatures(sc.initial_features())
    fr = list(sc.frames(steps + 20))
    g.upload_measurements(np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]), np.stack([f[2] for f in fr]))
    g.run_uploaded(0, 20, sc.dt)
    g.synchronize()
    t = time.perf_counter()
    g.run_uploaded(20, steps, sc.dt)
    g.synchronize()
    el = time.perf_counter() - t
    g.profile(True)
    g.run_uploaded(0, 10, sc.dt)
    g.synchronize()
    rep = g.profile_report()
    g.profile(False)
    n, m = 22 + 3 * N, 2 * N
    flops = 4.0 * n * n * m + 2.0 * n * m * m + m ** 3 / 3.0
    out["N=%d" % N] = {"steps_per_s": steps / el, "us_per_step": 1e6 * el / steps, "dense_form_update_TFLOPs": flops * steps / el / 1e12,
                       "stage_us": {k: 1e3 * v["ms"] / 10 for k, v in rep.items() if v["launches"]}}
    print("N=%d" % N, json.dumps(out["N=%d" % N]), flush=True)
    g.close()

# KLT front end: 640x480 pair, 64 and 256 points
IMG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "images")
a = np.asarray(Image.open(os.path.join(IMG, "640_480_test_gray.png")))
b = np.asarray(Image.open(os.path.join(IMG, "640_480_moved_test_gray.png")))
K = np.array([500.0, 0, 320.0, 0, 500.0, 240.0, 0, 0, 1.0], np.float32)
for npts in (64, 256):
    g = TightlyCoupledEKF(max_features=npts)
    t = KLTTracker(g)
    side = int(np.sqrt(npts))
    xs, ys = np.linspace(80, 560, side), np.linspace(60, 420, side)
    pts = np.array([[x, y] for y in ys for x in xs], np.float32)
    t.push_frame(a, K), t.push_frame(b, K)
    t.track_points(pts, pts.copy())
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        t.push_frame(b, K)
    t_push = (time.perf_counter() - t0) / reps
    t.push_frame(a, K), t.push_frame(b, K)
    t0 = time.perf_counter()
    for _ in range(reps):
        t.track_points(pts, pts.copy())
    t_track = (time.perf_counter() - t0) / reps
    g.profile(True)
    t.push_frame(a, K), t.push_frame(b, K)
    t.track_points(pts, pts.copy())
    rep = g.profile_report()
    g.profile(False)
    out["klt_%d" % npts] = {"push_frame_host_us": 1e6 * t_push, "track_host_us": 1e6 * t_track,
                            "pyramid_kernels_us": 1e3 * rep["klt_pyramid"]["ms"] / max(rep["klt_pyramid"]["launches"], 1),
                            "track_kernel_us": 1e3 * rep["klt_track"]["ms"] / max(rep["klt_track"]["launches"], 1)}
    print("klt %d" % npts, json.dumps(out["klt_%d" % npts]), flush=True)
    g.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/configs.json", "w"), indent=1)
