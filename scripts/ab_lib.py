"""Same-box A/B of two builds of the library: loads each .so given on the command line through ctypes (only entry points
both rounds have), runs the N=256 closed loop from the device-resident sequence and prints steps/s (best of 3) and a hash
of the final covariance (builds that only re-schedule must agree bit for bit)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd.sim import Scenario
from ekf_vio_amd.capi import Config

N, steps = 256, 200
sc = Scenario(N, seed=0)
uv = sc.initial_features()
fr = list(sc.frames(20 + steps))
z = np.ascontiguousarray(np.stack([f[0] for f in fr]), np.float32)
R = np.ascontiguousarray(np.stack([f[1] for f in fr]), np.float32)
p = np.ascontiguousarray(np.stack([f[2] for f in fr]), np.uint8)
fp, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
for path in sys.argv[1:]:
    lib = C.CDLL(path)
    cfg = Config()
    lib.ekfvio_default_config(C.byref(cfg))
    cfg.max_features = N
    h = C.c_void_p()
    lib.ekfvio_create.argtypes = [C.POINTER(Config), C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    assert lib.ekfvio_create(C.byref(cfg), 0, None, C.byref(h)) == 0
    lib.ekfvio_add_features.argtypes = [C.c_void_p, fp, C.c_int32]
    lib.ekfvio_upload_measurements.argtypes = [C.c_void_p, C.c_int32, fp, fp, u8p]
    lib.ekfvio_run_uploaded.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float]
    lib.ekfvio_synchronize.argtypes = [C.c_void_p]
    lib.ekfvio_reset.argtypes = [C.c_void_p]
    lib.ekfvio_destroy.argtypes = [C.c_void_p]
    best = 0.0
    for rep in range(3):
        lib.ekfvio_reset(h)
        assert lib.ekfvio_add_features(h, uv.ctypes.data_as(fp), N) == 0
        assert lib.ekfvio_upload_measurements(h, len(fr), z.ctypes.data_as(fp), R.ctypes.data_as(fp), p.ctypes.data_as(u8p)) == 0
        lib.ekfvio_run_uploaded(h, 0, 0, C.c_float(sc.dt))
        lib.ekfvio_run_uploaded(h, 0, 20, C.c_float(sc.dt))
        lib.ekfvio_synchronize(h)
        t0 = time.perf_counter()
        lib.ekfvio_run_uploaded(h, 20, steps, C.c_float(sc.dt))
        lib.ekfvio_synchronize(h)
        best = max(best, steps / (time.perf_counter() - t0))
    n = 22 + 3 * N
    sig = np.zeros((n, n), np.float32)
    lib.ekfvio_get_sigma.argtypes = [C.c_void_p, fp, C.c_int32]
    assert lib.ekfvio_get_sigma(h, sig.ctypes.data_as(fp), n) == 0
    import hashlib
    print("%-40s %8.1f steps/s  (%.2f us/step)  Sigma sha1 %s" % (os.path.basename(path), best, 1e6 / best, hashlib.sha1(sig.tobytes()).hexdigest()[:12]), flush=True)
    lib.ekfvio_destroy(h)
