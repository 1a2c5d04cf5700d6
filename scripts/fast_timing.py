"""Times frame ingest (resize + pyramid) and landmark replenishment (FAST + selection) on the reference's test image."""
import os, sys, time
import numpy as np
from PIL import Image
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import EKFVIO, capi  # noqa: E402

IMG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "images")
img = np.asarray(Image.open(os.path.join(IMG, "640_480_test_gray.png")))
K = np.array([500.0, 0, 320.0, 0, 500.0, 240.0, 0, 0, 1.0], np.float32)
for nf in (100, 256):
    v = EKFVIO(max_features=nf)
    v.addFrame(1.0, img, K)
    t0 = time.perf_counter()
    for _ in range(20):
        v.fast(50, True)
    t1 = time.perf_counter()
    px = v.replenishFeatures()
    t2 = time.perf_counter()
    print("max_features %d: cv::FAST equivalent incl. D2H of the list %.3f ms/call; replenishFeatures (FAST + first fit + addNewFeatures) %.3f ms, %d landmarks added"
          % (nf, 1e3 * (t1 - t0) / 20, 1e3 * (t2 - t1), len(px)))
    v.tc_ekf.close()
