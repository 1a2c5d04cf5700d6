import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
for b in (1, 2, 8):
    r = bench.concurrent_sequences(256, 0, b, 200)
    print(os.environ.get("EKFVIO_PERSIST_GAIN", "1"), b, "total %.0f steps/s" % r["total_steps_per_s"], flush=True)
