"""ctypes binding of libekfvio_hip.so (include/ekfvio.h).  No fallback: a missing or
unloadable library raises; nothing in this package computes on the CPU.

Two builds of the same sources (ekf_vio_amd/_build.py): the product library, which exports exactly what include/ekfvio.h
declares, and libekfvio_hip_hooks.so, which adds include/ekfvio_test_hooks.h (raw kernels, in-kernel stamps, fault
injection: tests and profiling scripts only).  load() is the product; load(hooks=True) the other."""
import ctypes as C
import os

from . import _build

OK, EINVAL, ENUMERIC, ECAPACITY, EDEVICE, ESTATE, EABORTED = range(7)
_ERR = {EINVAL: "EKFVIO_EINVAL", ENUMERIC: "EKFVIO_ENUMERIC", ECAPACITY: "EKFVIO_ECAPACITY",
        EDEVICE: "EKFVIO_EDEVICE", ESTATE: "EKFVIO_ESTATE", EABORTED: "EKFVIO_EABORTED"}
PREDICT_STRUCTURED, PREDICT_DENSE = 0, 1


class Config(C.Structure):
    _fields_ = [("max_features", C.c_int32), ("default_point_depth", C.c_float),
                ("default_point_depth_variance", C.c_float), ("default_point_homogenous_variance", C.c_float),
                ("predict_mode", C.c_int32), ("klt_window_size", C.c_int32), ("klt_max_pyramid_level", C.c_int32),
                ("klt_max_iterations", C.c_int32), ("klt_epsilon", C.c_float), ("klt_min_eigen", C.c_float),
                ("kill_pad", C.c_int32), ("max_image_width", C.c_int32), ("max_image_height", C.c_int32),
                ("use_principal_point", C.c_int32), ("inverse_image_scale", C.c_int32), ("fast_threshold", C.c_int32),
                ("min_new_feature_dist", C.c_int32), ("fast_blur_sigma", C.c_float), ("replenish", C.c_int32),
                ("sample_based_uncertainty", C.c_int32), ("use_imu", C.c_int32), ("imu_gyro_variance", C.c_float),
                ("imu_accel_variance", C.c_float), ("gravity", C.c_float * 3)]


class EkfvioError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("%s %s" % (_ERR.get(code, str(code)), msg))
        self.code = code


# every symbol include/ekfvio.h declares
SYMBOLS = ["ekfvio_default_config", "ekfvio_create", "ekfvio_destroy", "ekfvio_reset", "ekfvio_last_error",
           "ekfvio_add_features", "ekfvio_process", "ekfvio_linearize", "ekfvio_update", "ekfvio_measurement_map",
           "ekfvio_num_features", "ekfvio_dim", "ekfvio_get_base_mu", "ekfvio_get_features", "ekfvio_get_sigma",
           "ekfvio_get_feature_cov", "ekfvio_get_depth_variance", "ekfvio_set_feature_cov", "ekfvio_metric2pixel_map",
           "ekfvio_pixel2metric_map", "ekfvio_get_odometry", "ekfvio_get_points", "ekfvio_check_sigma", "ekfvio_set_state",
           "ekfvio_klt_push_frame", "ekfvio_klt_track", "ekfvio_klt_track_points", "ekfvio_klt_get_level",
           "ekfvio_klt_uncertainty_points", "ekfvio_step_image", "ekfvio_replenish", "ekfvio_fast_detect", "ekfvio_imu", "ekfvio_imu_update",
           "ekfvio_upload_measurements", "ekfvio_run_uploaded", "ekfvio_synchronize", "ekfvio_profile_enable",
           "ekfvio_profile_reset", "ekfvio_profile_count", "ekfvio_profile_name", "ekfvio_profile_get",
           "ekfvio_profile_update_gemms", "ekfvio_get_counters"]
# every symbol include/ekfvio_test_hooks.h declares (libekfvio_hip_hooks.so only)
HOOK_SYMBOLS = ["ekfvio_test_klt_padded_level", "ekfvio_test_blurred_level0", "ekfvio_test_gemm", "ekfvio_test_gemm_bench", "ekfvio_test_potrf_stamps",
                "ekfvio_test_sweep_stamps", "ekfvio_test_sweep_fault", "ekfvio_test_cholesky_solve"]

_libs = {}


def lib_path():
    return _build.LIB_PATH


def load(build_if_missing=True, hooks=False):
    """Loads the HIP library (building it with hipcc first if it is absent); hooks=True: the build with the test hooks."""
    if hooks in _libs:
        return _libs[hooks]
    path = _build.HOOKS_LIB_PATH if hooks else _build.LIB_PATH
    if build_if_missing:
        _build.build(hooks=hooks)  # (the hooks build only when asked for) staleness is decided under the build lock; hipcc cross-compiles gfx950 with or without a GPU
    elif not os.path.exists(path):
        raise FileNotFoundError(path + " not built; run python -m ekf_vio_amd._build")
    lib = C.CDLL(path)
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    fp, u8p, ip = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
    sig = {
        "ekfvio_default_config": [C.POINTER(Config)],
        "ekfvio_create": [C.POINTER(Config), C.c_int, vp, C.POINTER(vp)],
        "ekfvio_destroy": [vp], "ekfvio_reset": [vp],
        "ekfvio_add_features": [vp, fp, i32], "ekfvio_process": [vp, f32], "ekfvio_linearize": [vp, f32, fp],
        "ekfvio_update": [vp, fp, fp, u8p, i32], "ekfvio_measurement_map": [vp, u8p, i32, ip, ip],
        "ekfvio_num_features": [vp], "ekfvio_dim": [vp], "ekfvio_get_base_mu": [vp, fp],
        "ekfvio_get_features": [vp, fp, fp, u8p], "ekfvio_get_sigma": [vp, fp, i32],
        "ekfvio_get_feature_cov": [vp, i32, fp], "ekfvio_get_depth_variance": [vp, i32, fp],
        "ekfvio_set_feature_cov": [vp, i32, fp], "ekfvio_metric2pixel_map": [fp, fp], "ekfvio_pixel2metric_map": [fp, fp],
        "ekfvio_get_odometry": [vp, fp, fp, fp, fp], "ekfvio_get_points": [vp, fp, fp],
        "ekfvio_check_sigma": [vp, fp, fp], "ekfvio_set_state": [vp, i32, fp, fp, fp, u8p, fp, i32],
        "ekfvio_klt_push_frame": [vp, u8p, i32, i32, i32, fp], "ekfvio_klt_track": [vp, fp, fp, u8p],
        "ekfvio_klt_track_points": [vp, fp, fp, i32, fp, u8p],
        "ekfvio_klt_uncertainty_points": [vp, fp, fp, i32, fp],
        "ekfvio_klt_get_level": [vp, i32, ip, ip, u8p, C.POINTER(C.c_int16)],
        "ekfvio_step_image": [vp, C.c_double, u8p, i32, i32, i32, fp], "ekfvio_imu": [vp, C.c_double, fp, fp], "ekfvio_imu_update": [vp, fp, fp],
        "ekfvio_replenish": [vp, ip, ip], "ekfvio_fast_detect": [vp, i32, i32, i32, ip, ip, ip],
        "ekfvio_upload_measurements": [vp, i32, fp, fp, u8p], "ekfvio_run_uploaded": [vp, i32, i32, f32],
        "ekfvio_synchronize": [vp], "ekfvio_profile_enable": [vp, i32], "ekfvio_profile_reset": [vp],
        "ekfvio_profile_count": [], "ekfvio_profile_name": [i32],
        "ekfvio_profile_get": [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)],
        "ekfvio_profile_update_gemms": [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_double)],
        "ekfvio_get_counters": [vp, C.POINTER(C.c_int64)],
    }
    if hooks:
        sig.update({
            "ekfvio_test_klt_padded_level": [vp, i32, ip, u8p, C.POINTER(C.c_int16)],
            "ekfvio_test_blurred_level0": [vp, u8p],
            "ekfvio_test_gemm": [vp, i32, i32, i32, i32, f32, fp, i32, fp, i32, f32, fp, i32, i32],
            "ekfvio_test_cholesky_solve": [vp, i32, i32, fp, fp, fp, fp, ip],
            "ekfvio_test_potrf_stamps": [vp, C.POINTER(C.c_int64)],
            "ekfvio_test_sweep_stamps": [vp, C.c_int, C.POINTER(C.c_int64)],
            "ekfvio_test_sweep_fault": [vp, i32, i32],
            "ekfvio_test_gemm_bench": [vp, i32, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_double)],
        })
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.ekfvio_last_error.argtypes = [vp]
    lib.ekfvio_last_error.restype = C.c_char_p
    lib.ekfvio_profile_name.restype = C.c_char_p
    _libs[hooks] = lib
    return lib
