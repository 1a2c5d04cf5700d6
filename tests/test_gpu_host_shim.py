"""The C++ host shim (ekf_vio_amd/host/ekfvio.hpp) and the ROS-free replay driver run the
synthetic closed loop through the C-ABI from compiled C++ code, no Python in the loop."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu


def test_cpp_replay_driver_tracks_truth():
    from ekf_vio_amd import _build
    _build.build()
    exe = _build.build_host()
    out = subprocess.run([exe, "64", "120", "0"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"est_pos (\S+) (\S+) (\S+) truth_pos (\S+) (\S+) (\S+)", out.stdout)
    est, tru = [float(x) for x in m.groups()[:3]], [float(x) for x in m.groups()[3:]]
    assert max(abs(a - b) for a, b in zip(est, tru)) < 0.02
    assert "numeric_ok 1" in out.stdout
    md = float(re.search(r"min_diag (\S+)", out.stdout).group(1))
    assert md >= 0


def test_cpp_replay_of_records_equals_the_python_node(tmp_path):
    """ekfvio_replay --records: (stamp, image, K) and (stamp, imu) records in, odometry and point-cloud records out
    (last row), through ekfvio::EKFVIO::addFrame with cfg.replenish = 1.  The same records through the Python
    mirror of the node must give the same bits: both are the same C-ABI calls."""
    import numpy as np
    from PIL import Image
    from ekf_vio_amd import EKFVIO, _build, capi
    from ekf_vio_amd.sim import translated_sequence
    _build.build()
    exe = _build.build_host()
    base = np.asarray(Image.open(os.path.join(os.path.dirname(__file__), "golden", "images", "640_480_test_gray.png")))
    imgs = translated_sequence(base, 6)
    K = [500.0, 0.0, 320.0, 0.0, 500.0, 240.0, 0.0, 0.0, 1.0]
    lines = []
    for i, im in enumerate(imgs):
        name = "frame_%03d.pgm" % i
        with open(tmp_path / name, "wb") as fh:
            fh.write(b"P5\n# replay test\n640 480\n255\n" + im.tobytes())
        stamp = float("%.9f" % (100.0 + i / 30.0))  # what the text record holds
        lines.append("image %.9f %s %s" % (stamp, name, " ".join("%.9g" % k for k in K)))
        for j in range(3):  # IMU at a higher rate between the frames
            lines.append("imu %.9f 0.0 0.1 0.0 0.0 0.0 9.81" % (stamp + (j + 1) / 120.0))
    (tmp_path / "records.txt").write_text("# arrival order\n" + "\n".join(lines) + "\n")
    (tmp_path / "params.yaml").write_text("num_features: 48\ninverse_image_scale: 1\nfast_threshold: 50\n")
    out = subprocess.run([exe, "--records", str(tmp_path), "--params", str(tmp_path / "params.yaml")], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "6 images, 18 imu; 48 landmarks" in out.stdout
    odom = np.loadtxt(tmp_path / "odom.txt")
    assert odom.shape == (6, 15)
    clouds, cur = [], None
    for ln in (tmp_path / "points.txt").read_text().splitlines():
        if ln.startswith("cloud"):
            cur = []
            clouds.append((float(ln.split()[1]), int(ln.split()[2]), cur))
        else:
            cur.append([float(x) for x in ln.split()])
    assert len(clouds) == 6 and all(n == len(pts) for _, n, pts in clouds)
    # the Python mirror of the node on the same records
    v = EKFVIO(max_features=48, replenish=1, fast_threshold=50, inverse_image_scale=1)
    for i, im in enumerate(imgs):
        stamp = float("%.9f" % (100.0 + i / 30.0))
        rc = v.addFrame(stamp, im, np.array(K, np.float32))
        od = v.odometry()
        row = np.concatenate([od["position"], od["orientation_wxyz"], od["linear"], od["angular"]]).astype(np.float32)
        assert np.array_equal(odom[i, 1:14].astype(np.float32), row), i
        assert int(odom[i, 14]) == int(rc == capi.OK) and abs(odom[i, 0] - stamp) < 1e-6
        xyz, inten = v.points()
        got = np.array(clouds[i][2], np.float32).reshape(-1, 4)
        assert got.shape[0] == xyz.shape[0] == 48
        assert np.array_equal(got[:, :3], xyz) and np.array_equal(got[:, 3], inten), i
    assert np.isfinite(odom).all() and abs(np.linalg.norm(odom[-1, 4:8]) - 1) < 1e-5
    v.tc_ekf.close()


def _write_records(tmp_path, imgs, K, t0, extra_lines_after_image):
    lines = []
    for i, im in enumerate(imgs):
        name = "frame_%03d.pgm" % i
        h, w = im.shape
        with open(tmp_path / name, "wb") as fh:
            fh.write(b"P5\n%d %d\n255\n" % (w, h) + im.tobytes())
        stamp = float("%.9f" % (t0 + i / 30.0))
        lines.extend(extra_lines_after_image(i, stamp, before=True))
        lines.append("image %.9f %s %s" % (stamp, name, " ".join("%.9g" % k for k in K)))
        lines.extend(extra_lines_after_image(i, stamp, before=False))
    (tmp_path / "records.txt").write_text("\n".join(lines) + "\n")


def test_replay_with_imu_records_out_of_order_and_late(tmp_path):
    """with imu_update on, IMU records stamped AFTER an image reach the node before that image (200 Hz
    IMU, slower image transport), and a record can arrive after the frame that should have followed it.  The first kind is
    queued and applied in stamp order in front of the next frame, the second is dropped and counted; neither may abort the
    replay or lose a frame.  The Python node on the same arrival order must give the same bits."""
    import numpy as np
    from PIL import Image
    from ekf_vio_amd import EKFVIO, _build, capi
    from ekf_vio_amd.sim import translated_sequence
    _build.build()
    exe = _build.build_host()
    base = np.asarray(Image.open(os.path.join(os.path.dirname(__file__), "golden", "images", "640_480_test_gray.png")))
    imgs = translated_sequence(base, 5)
    K = [500.0, 0.0, 320.0, 0.0, 500.0, 240.0, 0.0, 0.0, 1.0]
    arrivals = []  # what both nodes see, in arrival order

    def extras(i, stamp, before):
        if before:
            # records stamped up to 12 ms AFTER this image, delivered before it (and out of order among themselves)
            stamps = [stamp + 0.012, stamp + 0.004, stamp + 0.008]
        else:
            stamps = [stamp + 0.016] + ([stamp - 0.010] if i == 2 else [])  # the second: older than the frame just processed
        lines = []
        for t in stamps:
            arrivals.append(("imu", float("%.9f" % t)))
            lines.append("imu %.9f 0.0 0.02 0.0 0.0 -9.81 0.0" % t)
        if before:
            arrivals.append(("image", i, stamp))  # the image record follows the `before` batch
        return lines

    _write_records(tmp_path, imgs, K, 50.0, extras)
    (tmp_path / "params.yaml").write_text("num_features: 40\ninverse_image_scale: 1\nimu_update: true\n")
    out = subprocess.run([exe, "--records", str(tmp_path), "--params", str(tmp_path / "params.yaml")], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "5 images, 21 imu" in out.stdout
    assert "imu: 1 records dropped" in out.stdout, out.stdout
    odom = np.loadtxt(tmp_path / "odom.txt")
    assert odom.shape == (5, 15) and np.isfinite(odom).all()
    v = EKFVIO(max_features=40, replenish=1, inverse_image_scale=1, use_imu=1)
    row = 0
    for rec in arrivals:
        if rec[0] == "imu":
            v.imu_callback(rec[1], [0.0, 0.02, 0.0], [0.0, -9.81, 0.0])
        else:
            rc = v.addFrame(rec[2], imgs[rec[1]], np.array(K, np.float32))
            od = v.odometry()
            want = np.concatenate([od["position"], od["orientation_wxyz"], od["linear"], od["angular"]]).astype(np.float32)
            assert np.array_equal(odom[row, 1:14].astype(np.float32), want), row
            assert int(odom[row, 14]) == int(rc == capi.OK)
            row += 1
    assert row == 5 and v.dropped_imu == 1
    v.tc_ekf.close()


def test_replay_accepts_a_1280x960_sequence_and_writes_insight(tmp_path):
    """no image size in any configuration; the node's default scale 4.  Also publishInsight's image
    (EKFVIO.cpp:379-442) from the C++ shim against the Python mirror's."""
    import numpy as np
    from PIL import Image
    from ekf_vio_amd import EKFVIO, _build
    from ekf_vio_amd.sim import translated_sequence
    from test_gpu_loop import big_image
    _build.build()
    exe = _build.build_host()
    imgs = translated_sequence(big_image(1280, 960), 3, dx=-4.0, dy=-2.0)
    K = [900.0, 0.0, 640.0, 0.0, 900.0, 480.0, 0.0, 0.0, 1.0]
    _write_records(tmp_path, imgs, K, 9.0, lambda i, stamp, before: [])
    (tmp_path / "params.yaml").write_text("num_features: 60\n")  # inverse_image_scale: the node's default 4
    out = subprocess.run([exe, "--records", str(tmp_path), "--params", str(tmp_path / "params.yaml"), "--insight"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "3 images, 0 imu" in out.stdout
    v = EKFVIO(max_features=60, replenish=1, inverse_image_scale=4)
    for i, im in enumerate(imgs):
        v.addFrame(float("%.9f" % (9.0 + i / 30.0)), im, np.array(K, np.float32))
        got = np.asarray(Image.open(tmp_path / ("insight_%03d.ppm" % i)))  # RGB on disk
        want = v.insight()[:, :, ::-1]
        assert got.shape == (240, 320, 3) and np.array_equal(got, want), i
    green = (want[:, :, 1] == 255) & (want[:, :, 0] == 0) & (want[:, :, 2] == 0)
    live = int((v.tc_ekf.get_state()["del_flag"] == 0).sum())
    assert live > 10 and green.sum() > 40 * live / 2  # a square outline of 88 pixels per live landmark, overlaps allowed
    v.tc_ekf.close()
