"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads without a GPU
and exports every symbol include/ekfvio.h declares; nothing in the product package imports
the oracle."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from ekf_vio_amd import capi
    lib = capi.load()
    hdr = open(os.path.join(ROOT, "include", "ekfvio.h")).read()
    declared = set(re.findall(r"\b(ekfvio_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.lib_path()]).decode()
    exported = set(re.findall(r"\bT (ekfvio_[a-z0-9_]+)", out))
    assert declared <= exported


def test_default_config_matches_reference_params():
    """Params.h D_* defaults the hot path reads (Params.h:33,36,46,83-86,103-104)."""
    import ctypes as C
    from ekf_vio_amd import capi
    lib = capi.load()
    cfg = capi.Config()
    assert lib.ekfvio_default_config(C.byref(cfg)) == capi.OK
    assert cfg.max_features == 100 and cfg.kill_pad == 11
    assert abs(cfg.default_point_depth - 0.5) < 1e-7 and cfg.default_point_depth_variance == 100.0
    assert abs(cfg.default_point_homogenous_variance - 1e-5) < 1e-12
    assert cfg.klt_window_size == 21 and cfg.klt_max_pyramid_level == 3 and cfg.klt_max_iterations == 30
    assert abs(cfg.klt_min_eigen - 1e-4) < 1e-10 and abs(cfg.klt_epsilon - 0.01) < 1e-9
    assert lib.ekfvio_default_config(None) == capi.EINVAL


def test_null_handle_is_rejected_not_crashed():
    from ekf_vio_amd import capi
    lib = capi.load()
    assert lib.ekfvio_process(None, 0.1) == capi.EINVAL
    assert lib.ekfvio_destroy(None) == capi.EINVAL
    assert lib.ekfvio_num_features(None) == -1


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: no product source may import, link or call it."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "ekf_vio_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(base, fn), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M) or "ekf_oracle" in txt or "libekf_oracle" in txt:
                    bad.append(fn)
    assert not bad, bad
    code = "import sys; sys.path.insert(0, %r); import ekf_vio_amd; assert 'oracle' not in sys.modules" % ROOT
    subprocess.check_call([sys.executable, "-c", code])


def test_create_fails_loudly_without_gpu(gpu_available):
    """No CPU fallback: without a device the handle cannot be created."""
    if gpu_available:
        return
    import pytest
    from ekf_vio_amd import EkfvioError, TightlyCoupledEKF
    with pytest.raises(EkfvioError):
        TightlyCoupledEKF(max_features=4)
