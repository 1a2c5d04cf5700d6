"""In-process A/B of an environment knob on the image loop (the knobs are read when a handle is created): alternating handles,
medians over repetitions.  usage: loop_ab.py KNOB A B [frames]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
knob, A, B = sys.argv[1], sys.argv[2], sys.argv[3]
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 150
res = {}
for rep in range(4):
    for v in (A, B):
        os.environ[knob] = v
        for name, kw in (("n256", dict(n_landmarks=256)), ("n64", dict(n_landmarks=64)), ("node+outputs", dict(n_landmarks=100, node_defaults=True, outputs=True))):
            r = bench.full_loop(device=0, frames=frames, **kw)
            res.setdefault((name, v), []).append(1e3 * r["ms_per_frame"])
for (name, v), xs in sorted(res.items()):
    print("%-14s %s=%s  median %.1f us/frame  (%s)" % (name, knob, v, float(np.median(xs)), " ".join("%.1f" % x for x in xs)))
