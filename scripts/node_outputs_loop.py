import os, sys
sys.path.insert(0, os.getcwd())
import bench
r = bench.full_loop(100, 0, node_defaults=True, outputs=True)
print(round(r["frames_per_s"]), round(1e3 * r["ms_per_frame"], 1), r["landmarks"])
