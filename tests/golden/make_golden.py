"""Regenerates tests/golden/kat.json.

Sources of the expected values:
  * h_map: the reference's own known-answer test (test/test_ekf.cpp:44-63).
  * sigma0_diag / q_diag_dt0p1: constants of TightlyCoupledEKF.cpp:29-54,87-91,126-171.
  * scenarios A-E: inputs of test/test_ekf.cpp:154-204 (feature (0.1,0.1) at default depth
    0.5 -> rho=2); expected outputs evaluated in fp64 from the reference formulas
    (TightlyCoupledEKF.cpp:328-460) by the independent numpy restatement
    oracle/np_oracle.py and cross-checked against SURVEY.md section 8(c).
The reference binary cannot be run here (no ROS/Eigen/OpenCV), so these are derived known
answers, not recorded outputs.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle.np_oracle import convolve_base_state, convolve_feature  # noqa: E402


def main():
    out = {}
    out["h_map"] = {"n_features": 3, "measured": [1, 0, 1], "rows": 4, "cols": 31,
                    "ones_at": [[0, 22], [1, 23], [2, 28], [3, 29]]}
    out["sigma0_diag_3feat"] = [0.0] * 7 + [30.0] * 9 + [0.5] * 6 + [1e-5, 1e-5, 100.0] * 3
    out["q_diag_dt0p1_3feat"] = [1e-5] * 7 + [1e-3] * 3 + [0.5] * 6 + [1e-4] * 6 + [1e-5] * 9
    mu = np.zeros(22)
    mu[3] = 1.0
    feat = np.array([0.1, 0.1, 2.0])
    scen = []

    def add(name, mu, feat, dt):
        scen.append({"name": name, "base_mu": mu.tolist(), "feature": feat.tolist(), "dt": dt,
                     "base_out": convolve_base_state(mu, dt).tolist(),
                     "feature_out": convolve_feature(mu, feat, dt).tolist()})

    add("A_init", mu.copy(), feat, 0.1)
    mu[9] = 1.0
    add("B_bdz1", mu.copy(), feat, 0.1)
    mu[10] = 3.14
    add("C_omx", mu.copy(), feat, 0.1)
    mu[10] = 0.0
    mu[12] = 3.14
    add("D_omz", mu.copy(), feat, 0.1)
    mu[12] = 0.0
    mu[11] = -3.1415
    mu[9] = 1.0
    mu[7] = 1.0
    add("E_omy", mu.copy(), np.array([0.0, 0.0, 2.0]), 0.5)
    out["scenarios"] = scen
    # spot values quoted in SURVEY.md 8(c) (7 significant digits) to pin the generator itself
    out["survey_spot"] = {
        "C_quat": [0.9877008, 0.1563558, 0.0, 0.0], "C_vel": [0.0, 0.3088655, 0.9511057],
        "C_feature": [0.1369867, 0.4687725, 2.7397334], "D_feature": [0.1574964, 0.0802800, 2.5],
        "E_pos": [0.5, 0.0, 0.5], "E_quat": [0.7071232, 0.0, -0.7070904, 0.0],
        "E_vel": [1.0000463, 0.0, -0.9999537], "E_feature": [-0.0000463, 0.0, 2.0], "B_feature": [0.125, 0.125, 2.5]}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
