"""Frames/s of the node's loop (ekfvio_step_image + odometry + point cloud per frame) at the node's defaults and at N = 256."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
out = {}
for name, kw in (("node_defaults_with_outputs", dict(n_landmarks=100, node_defaults=True, outputs=True)),
                 ("node_defaults", dict(n_landmarks=100, node_defaults=True)),
                 ("n256_with_outputs", dict(n_landmarks=256, outputs=True)), ("n256", dict(n_landmarks=256)),
                 ("n64_with_outputs", dict(n_landmarks=64, outputs=True)), ("n64", dict(n_landmarks=64))):
    r = bench.full_loop(device=0, **kw)
    out[name] = dict(frames_per_s=round(r["frames_per_s"], 1), us_per_frame=round(1e3 * r["ms_per_frame"], 1), landmarks=r["landmarks"])
    print(name, out[name], flush=True)
