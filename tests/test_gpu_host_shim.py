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
