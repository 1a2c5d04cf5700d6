"""CPU oracle for the EKF-VIO hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (ekf_vio_amd) never does.
"""
from .oracle_py import (KltFrame, OracleFilter, amd_order, build_oracle, circle_fill, fast_detect, frame_resize, gauss5_kernel, gaussian_blur5,  # noqa: F401
                        klt_track, klt_uncertainty, max_threads, oracle_lib, replenish, set_threads)
