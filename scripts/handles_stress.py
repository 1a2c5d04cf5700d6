"""Eight handles with their device-resident runs in flight together against the same sequences run alone, repeated:
counts the repetitions in which any handle's final covariance differs (a data race that only concurrency exposes).
usage: python scripts/handles_stress.py <lib.so> [reps] [N]"""
import ctypes as C, hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd.sim import Scenario
from ekf_vio_amd.capi import Config

path = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N = int(sys.argv[3]) if len(sys.argv) > 3 else 256
steps, H = 12, 8
lib = C.CDLL(path)
fp, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
lib.ekfvio_create.argtypes = [C.POINTER(Config), C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
lib.ekfvio_add_features.argtypes = [C.c_void_p, fp, C.c_int32]
lib.ekfvio_upload_measurements.argtypes = [C.c_void_p, C.c_int32, fp, fp, u8p]
lib.ekfvio_run_uploaded.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float]
lib.ekfvio_synchronize.argtypes = [C.c_void_p]
lib.ekfvio_reset.argtypes = [C.c_void_p]
lib.ekfvio_get_sigma.argtypes = [C.c_void_p, fp, C.c_int32]
n = 22 + 3 * N
seqs = []
for seed in range(H):
    sc = Scenario(N, seed=seed)
    fr = list(sc.frames(steps))
    seqs.append((sc, sc.initial_features(), [np.ascontiguousarray(np.stack([f[i] for f in fr]), t) for i, t in ((0, np.float32), (1, np.float32), (2, np.uint8))]))
hs = []
for _ in range(H):
    cfg = Config()
    lib.ekfvio_default_config(C.byref(cfg))
    cfg.max_features = N
    h = C.c_void_p()
    assert lib.ekfvio_create(C.byref(cfg), 0, None, C.byref(h)) == 0
    hs.append(h)

def prepare(h, s):
    sc, uv, (z, R, p) = s
    lib.ekfvio_reset(h)
    assert lib.ekfvio_add_features(h, uv.ctypes.data_as(fp), N) == 0
    assert lib.ekfvio_upload_measurements(h, steps, z.ctypes.data_as(fp), R.ctypes.data_as(fp), p.ctypes.data_as(u8p)) == 0
    lib.ekfvio_run_uploaded(h, 0, 0, C.c_float(sc.dt))

def digest(h):
    sig = np.zeros((n, n), np.float32)
    assert lib.ekfvio_get_sigma(h, sig.ctypes.data_as(fp), n) == 0
    return hashlib.sha1(sig.tobytes()).hexdigest()[:12]

solo = []
for h, s in zip(hs, seqs):
    prepare(h, s)
    lib.ekfvio_run_uploaded(h, 0, steps, C.c_float(s[0].dt))
    lib.ekfvio_synchronize(h)
    solo.append(digest(h))
bad = 0
for rep in range(reps):
    for h, s in zip(hs, seqs):
        prepare(h, s)
    for h, s in zip(hs, seqs):
        lib.ekfvio_synchronize(h)
    for h, s in zip(hs, seqs):
        lib.ekfvio_run_uploaded(h, 0, steps, C.c_float(s[0].dt))
    for h in hs:
        lib.ekfvio_synchronize(h)
    got = [digest(h) for h in hs]
    diff = [i for i in range(H) if got[i] != solo[i]]
    if diff:
        bad += 1
        print("rep %d: handles %s differ" % (rep, diff), flush=True)
print("%s: %d of %d repetitions differed" % (os.path.basename(path), bad, reps))
