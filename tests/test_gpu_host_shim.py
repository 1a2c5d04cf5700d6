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
