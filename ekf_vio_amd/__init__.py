"""ekf_vio_amd — MI355X (gfx950) backend for the EKF-VIO per-frame hot path.

Layout: csrc/ (HIP kernels + the C-ABI of include/ekfvio.h), capi.py (ctypes binding),
filter.py (host mirror of the reference's TightlyCoupledEKF interface), sim.py (ROS-free
synthetic scenario generator).  There is no CPU compute path in this package.
"""
from .capi import EkfvioError, load  # noqa: F401
from .filter import EKFVIO, KLTTracker, TightlyCoupledEKF  # noqa: F401
