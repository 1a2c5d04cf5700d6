"""How many Gauss-Newton steps do the landmarks of the image loop take per pyramid level, and what decides the tracker launch's duration?
(VERDICT r04 next #6.)  Runs the N = 256 image loop (bench.full_loop's settings) with the kernels' diagnostic buffer on and reads, for a few frames,
every landmark's iteration count per level (klt_track_kernel packs a byte per level into dbg[landmark])."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from ekf_vio_amd import EKFVIO
from ekf_vio_amd.sim import translated_sequence
from test_gpu_loop import K, grey

seq = translated_sequence(grey(), 40)
v = EKFVIO(max_features=256, replenish=1, fast_threshold=20, min_new_feature_dist=12, hooks=True)
lib, h = v.tc_ekf.lib, v.tc_ekf.h
rows = []
for i, img in enumerate(seq):
    if i >= 6:
        lib.ekfvio_test_sweep_stamps(h, 1, None)
    v.addFrame(1.0 + i / 30.0, img, K)
    if i >= 6:
        buf = (C.c_int64 * 1024)()
        lib.ekfvio_test_sweep_stamps(h, 1, buf)
        n = v.tc_ekf.num_features
        p = np.array(list(buf)[:min(n, 896)], np.int64)
        # (the sweep's own stamps share the buffer: slots 0-3, 32-95, 500-507, 600-661, 700-749 belong to them)
        keep = np.array([q for q in range(len(p)) if not (q < 4 or 32 <= q < 96 or 500 <= q < 508 or 600 <= q < 662 or 700 <= q < 750)])
        p = p[keep]
        rows.append(np.stack([(p >> (8 * l)) & 255 for l in range(4)], 1))  # [landmark, level]
its = np.concatenate(rows)  # all landmarks of all recorded frames
tot = its.sum(1)
print("landmark-frames recorded: %d (frames %d)" % (len(its), len(rows)))
for l in (3, 2, 1, 0):
    c = np.bincount(its[:, l], minlength=31)
    print("level %d: mean %.2f  median %d  p90 %d  max %d   histogram 0..30: %s" % (l, its[:, l].mean(), np.median(its[:, l]), np.percentile(its[:, l], 90), its[:, l].max(), c[:31].tolist()))
print("all four levels together: mean %.1f  median %d  p90 %d  p99 %d  max %d" % (tot.mean(), np.median(tot), np.percentile(tot, 90), np.percentile(tot, 99), tot.max()))
per_frame_max = [r.sum(1).max() for r in rows]
per_frame_mean = [r.sum(1).mean() for r in rows]
print("per frame: slowest landmark %s steps (mean of the frame's landmarks %.1f)" % ([int(x) for x in per_frame_max], float(np.mean(per_frame_mean))))
# the launch lasts as long as its slowest wavefront: 4 levels x (loads 0.75 + template 0.87 us) + steps x 0.67 us
print("estimated tracker launch: %.1f us with the slowest landmark (%.0f steps), %.1f us if every landmark took the mean (%.1f steps)"
      % (4 * 1.62 + 0.67 * np.mean(per_frame_max), np.mean(per_frame_max), 4 * 1.62 + 0.67 * np.mean(per_frame_mean), np.mean(per_frame_mean)))
