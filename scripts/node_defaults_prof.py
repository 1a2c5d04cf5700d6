"""The node's default configuration (N = 100, inverse_image_scale 4) frame loop, for rocprofv3: 60 frames of ekfvio_step_image."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
r = bench.full_loop(100, 0, frames=60, warm=6, node_defaults=True)
print({k: r[k] for k in ("frames_per_s", "ms_per_frame", "landmarks", "landmarks_never_lost", "stage_us_per_frame")})
