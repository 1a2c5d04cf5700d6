"""CPU oracle for the EKF-VIO hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (ekf_vio_amd) never does.
"""
from .oracle_py import OracleFilter, build_oracle, oracle_lib, set_threads, max_threads  # noqa: F401
