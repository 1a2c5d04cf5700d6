"""Soak run of the image loop (ekfvio_step_image with replenishment and per-frame outputs): 1500 frames at N = 256 and at the node defaults; the state stays finite."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
for N, frames in ((256, 1500), (100, 1500)):
    t = time.time()
    r = bench.full_loop(N, 0, frames=frames, warm=6, outputs=True, node_defaults=(N == 100))
    print(N, "frames", frames, "frames/s %.0f" % r["frames_per_s"], "landmarks", r["landmarks"], "never lost", r["landmarks_never_lost"], "numeric warnings", r["numeric_warnings"], "finite", r["state_finite"], "%.1f s" % (time.time() - t), flush=True)
