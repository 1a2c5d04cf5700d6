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
    # the product library exports EXACTLY the boundary: no test hook, no fault injector, and since round 6 no
    # launcher, kernel handle, template instance or data object either (-fvisibility=hidden, EKFVIO_API, csrc/ekfvio.map): the WHOLE
    # dynamic symbol table is compared with the header, whatever the symbols are called
    exported = {l.split()[-1] for l in out.splitlines() if l.strip()}
    assert declared == exported, declared ^ exported
    assert all(l.split()[-2] == "T" for l in out.splitlines() if l.strip()), out
    api_marked = set(re.findall(r"^EKFVIO_API [a-z \*]*?\b(ekfvio_[a-z0-9_]+)\s*\(", hdr, re.M))
    assert api_marked == declared, api_marked ^ declared


def test_hooks_build_adds_exactly_the_test_hooks_header():
    """libekfvio_hip_hooks.so = the same sources with -DEKFVIO_TEST_HOOKS: the boundary plus include/ekfvio_test_hooks.h (raw kernels,
    in-kernel stamps, fault injection), for the tests and the profiling scripts only."""
    from ekf_vio_amd import _build, capi
    lib = capi.load(hooks=True)
    hdr = open(os.path.join(ROOT, "include", "ekfvio_test_hooks.h")).read()
    hooks = set(re.findall(r"\b(ekfvio_test_[a-z0-9_]+)\s*\(", hdr))
    assert hooks and hooks == set(capi.HOOK_SYMBOLS), hooks ^ set(capi.HOOK_SYMBOLS)
    for s in hooks:
        assert hasattr(lib, s), s
    out = subprocess.check_output(["nm", "-D", "--defined-only", _build.HOOKS_LIB_PATH]).decode()
    exported = {l.split()[-1] for l in out.splitlines() if l.strip()}
    assert exported == hooks | set(capi.SYMBOLS), exported ^ (hooks | set(capi.SYMBOLS))
    assert "ekfvio_test" not in open(os.path.join(ROOT, "include", "ekfvio.h")).read()


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


def _print_config(tmp_path, text):
    import json
    from ekf_vio_amd import _build
    _build.build()
    exe = _build.build_host()
    args = [exe, "--print-config"]
    if text is not None:
        f = tmp_path / "params.yaml"
        f.write_text(text)
        args.append(str(f))
    out = subprocess.run(args, capture_output=True, text=True, timeout=120)
    return out.returncode, (json.loads(out.stdout) if out.returncode == 0 else out.stderr)


def test_node_parameter_names_map_onto_the_config(tmp_path):
    """the node's private ROS parameters (EKFVIO.cpp:19-67, defaults Params.h) reach ekfvio_config
    through the C++ shim's ekfvio::Params, by the reference's own names; no GPU involved."""
    rc, d = _print_config(tmp_path, None)
    assert rc == 0
    # Params.h defaults, incl. D_INVERSE_IMAGE_SCALE 4 (the C-ABI's own default is 1) and the node-level strings
    assert (d["max_features"], d["fast_threshold"], d["inverse_image_scale"], d["kill_pad"]) == (100, 50, 4, 11)
    assert d["min_new_feature_dist"] == 30 and d["klt_window_size"] == 21 and d["klt_max_pyramid_level"] == 3
    assert d["fast_blur_sigma"] == 0 and d["sample_based_uncertainty"] == 0
    assert d["node"]["odom_topic"] == "invio/odom" and d["node"]["camera_topic"] == "/camera/image_rect"
    assert d["node"]["imu_topic"] == "imu/measurement" and d["node"]["base_frame"] == "base_link"
    # a parameter file in the layout of the reference's params/*.yaml: hot-path names, names of subsystems the
    # reference no longer calls (accepted, ignored), comments, a namespaced key, a quoted string
    text = """
min_new_feature_dist: 25.0
num_features: 400
fast_threshold: 45   # corner threshold
fast_blur_sigma: 1.5
depth_translation_ratio: 0.01
default_point_depth: 0.75
default_point_depth_variance: 1000
minumum_depth_determinant: 0.00001
# how many points to update per frame
max_depth_updates_per_frame: 400
huber_width: 1e-6
moba_max_iterations: 10
inverse_image_scale: 2
publish_insight: false
/ekf_vio/kill_pad: 13
~min_klt_eigen_val: 0.001
max_pyramids: 2
klt_window_size: 15
odom_topic: "vio/odom"
frame_buffer_size: 2
"""
    rc, d = _print_config(tmp_path, text)
    assert rc == 0, d
    assert (d["max_features"], d["fast_threshold"], d["inverse_image_scale"], d["kill_pad"]) == (400, 45, 2, 13)
    assert d["min_new_feature_dist"] == 25 and d["fast_blur_sigma"] == 1.5
    assert d["default_point_depth"] == 0.75 and d["default_point_depth_variance"] == 1000
    assert abs(d["klt_min_eigen"] - 1e-3) < 1e-9 and d["klt_max_pyramid_level"] == 2 and d["klt_window_size"] == 15
    assert d["node"]["publish_insight"] == "false" and d["node"]["odom_topic"] == "vio/odom"
    # what the reference would silently accept but cannot mean anything here is an error, not a surprise
    for bad in ("num_feature: 10\n", "fast_threshold: many\n", "frame_buffer_size: 3\n", "inverse_image_scale: 2.5\n", "just words\n"):
        rc, err = _print_config(tmp_path, bad)
        assert rc == 2 and "error" in err, (bad, err)


def test_ros_node_source_type_checks_against_stub_headers():
    """the ROS1 front end (ekf_vio_amd/host/ros/ekfvio_node.cpp) cannot be built here (no ROS in the
    image or on the GPU boxes).  It is type-checked against declaration-only stand-ins for the ROS headers it includes
    (tests/ros_stubs), so that the node source and the host shim it calls stay in step; without ROS headers on the include
    path the translation unit is empty and must compile as such."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "ekf_vio_amd", "host", "ros", "ekfvio_node.cpp")
    base = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror"]
    out = subprocess.run(base + ["-I", os.path.join(root, "tests", "ros_stubs"), src], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    out = subprocess.run(base + [src], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    text = open(src).read()
    for needle in ("subscribeCamera", "advertise<nav_msgs::Odometry>", "advertise<sensor_msgs::PointCloud>", "waitForTransform",
                   "\"intensity\"", "imu_topic", "sendTransform"):
        assert needle in text, needle


def test_committed_pivot_chain_is_what_its_generator_emits():
    """ekf_vio_amd/csrc/potrf_chain.inc is generated (scripts/gen_potrf_chain.py) and compiled into the product: the
    committed file must be exactly what the generator emits today, so an edit of either without the other fails here."""
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "gen_potrf_chain.py")]).decode()
    have = open(os.path.join(ROOT, "ekf_vio_amd", "csrc", "potrf_chain.inc")).read()
    assert out == have


def test_chain_publication_wait_counts_the_loads_behind_the_stores():
    """chol_persist_kernel's chain raises ready[k] behind a COUNTED `s_waitcnt vmcnt(N)`: the wavefront's write-through stores
    of the factor must be older than exactly the N tile loads issued in front of that wait, and nothing else may sit between
    them.  The count is the compiler's to keep, so ekf_vio_amd/_build.py checks the ISA of the very compile it links (same
    flags) and refuses to build on a mismatch; here: the checker itself on good and bad ISA, and the stamp of the shipped
    build."""
    from ekf_vio_amd import _build
    good = "_ZN1x19chol_persist_kernelE:\n buffer_store_dwordx4 v[0:3], v4, s[0:3], 0 offen sc1\n" + " buffer_load_dword v1, v2, s[0:3], 0 offen\n" * 3 + \
           " ;;#ASMSTART\n s_waitcnt vmcnt(3)\n ;;#ASMEND\n codeLenInByte = 4\n"
    assert _build.check_counted_waits(good) == []
    assert _build.check_counted_waits(good.replace("vmcnt(3)", "vmcnt(2)")) != []
    assert _build.check_counted_waits(good.replace("sc1\n", "sc1\n buffer_load_dword v9, v2, s[0:3], 0 offen\n", 1)) != []
    assert _build.check_counted_waits("nothing here") != []
    # a kernel without ANY counted inline-asm wait must not pass (marker format changed, path restructured), and the
    # store in front of the wait must be a write-through one
    assert _build.check_counted_waits(good.replace("vmcnt(3)", "vmcnt(0)")) != []
    assert _build.check_counted_waits(good.replace(";;#ASMSTART", ";;#SOMETHING")) != []
    assert _build.check_counted_waits(good.replace(" offen sc1\n", " offen\n", 1)) != []
    # a library whose ISA stamp is missing counts as stale: build() re-checks instead of trusting it.  Both builds carry the kernel (the
    # hooks build is the one the abort, fault-injection and stamp tests run): both are checked
    for hooks in (False, True):
        v = _build._Variant(hooks)
        if os.path.exists(v.isa_stamp) and os.path.exists(v.lib):
            import fcntl
            with open(os.path.join(_build.LIB_DIR, ".build.lock"), "w") as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                os.rename(v.isa_stamp, v.isa_stamp + ".moved")
                try:
                    assert v.stale()
                finally:
                    os.rename(v.isa_stamp + ".moved", v.isa_stamp)
        _build.build(hooks=hooks)
        assert os.path.exists(v.isa_stamp), "the shipped library was linked without the ISA check"


def test_product_library_does_not_need_the_hooks_build():
    """a deployment that ships only libekfvio_hip.so must import and load without hipcc: the product's staleness does not look
    at the hooks library, which is built lazily by load(hooks=True)."""
    from ekf_vio_amd import _build
    _build.build()
    v = _build._Variant(False)
    hooks_lib = _build.HOOKS_LIB_PATH
    moved = os.path.exists(hooks_lib)
    import fcntl
    with open(os.path.join(_build.LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if moved:
            os.rename(hooks_lib, hooks_lib + ".moved")
        try:
            assert not v.stale()
            assert _build._Variant(True).stale()
        finally:
            if moved:
                os.rename(hooks_lib + ".moved", hooks_lib)
