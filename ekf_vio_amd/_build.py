"""Builds ekf_vio_amd/lib/libekfvio_hip.so from csrc/*.hip with hipcc for gfx950.

In-tree build on purpose: the .so travels with the repository snapshot to the GPU box.
"""
import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libekfvio_hip.so")
# the same sources with -DEKFVIO_TEST_HOOKS: the product library + include/ekfvio_test_hooks.h (raw kernels, stamps, fault injection).
# Tests and profiling scripts that need those entry points load THIS one; the product library exports none of them.
HOOKS_LIB_PATH = os.path.join(LIB_DIR, "libekfvio_hip_hooks.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
HOOKS_OBJ_DIR = os.path.join(LIB_DIR, "obj_hooks")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC",
         "-ffp-contract=off",  # the reference's x86-64 build never fuses multiply-add (parity)
         # the library's only dynamic symbols are the entry points include/ekfvio.h marks EKFVIO_API (tests/test_abi_cpu.py checks
         # the whole `nm -D --defined-only` set): no launcher, kernel stub or template instance can interpose in a node's process
         "-fvisibility=hidden", "-fvisibility-inlines-hidden",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value"]
# per-file extras.  chol.hip: keep MFMA results in VGPRs: its dependent MFMA -> MFMA chains feed each
# result straight back as an operand, and an AGPR destination costs a v_accvgpr_read per hop.
FILE_FLAGS = {"chol.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


FLAGS_STAMP = os.path.join(LIB_DIR, ".build.flags")
VERSION_SCRIPT = os.path.join(CSRC, "ekfvio.map")


def _extra_flags():
    return os.environ.get("EKFVIO_EXTRA_HIPCC_FLAGS", "").split()  # diagnostics, e.g. -DEKF_GEMM_STAMPS


def _flags_key():
    return " ".join(FLAGS + sorted("%s:%s" % kv for kv in ((k, " ".join(v)) for k, v in FILE_FLAGS.items())) + _extra_flags())


class _Variant:
    """One build of the sources: the product library, or the same sources with -DEKFVIO_TEST_HOOKS."""

    def __init__(self, hooks):
        self.hooks = hooks
        self.lib = HOOKS_LIB_PATH if hooks else LIB_PATH
        self.obj_dir = HOOKS_OBJ_DIR if hooks else OBJ_DIR
        self.defines = ["-DEKFVIO_TEST_HOOKS"] if hooks else []
        self.flags_stamp = FLAGS_STAMP + (".hooks" if hooks else "")
        self.isa_stamp = os.path.join(self.obj_dir, "chol.isa.checked")

    def deps(self):
        return (sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) +
                [VERSION_SCRIPT, os.path.join(_HERE, "..", "include", "ekfvio.h"), os.path.join(_HERE, "..", "include", "ekfvio_test_hooks.h"), __file__])

    def stale(self):
        """Only meaningful while holding the build lock (a concurrent builder replaces the library atomically, but the
        object files and the flags stamp change underneath)."""
        if not os.path.exists(self.lib) or not os.path.exists(self.flags_stamp) or not os.path.exists(self.isa_stamp):
            return True  # (no ISA stamp: the hand-placed counted wait of chol.hip has not been checked against this library's compile)
        if open(self.flags_stamp).read() != _flags_key():
            return True  # e.g. a diagnostic build with pricing switches (wrong results) must not survive
        t = os.path.getmtime(self.lib)
        return any(os.path.getmtime(d) > t for d in self.deps())


def build(force=False, verbose=False, hooks=False):
    """Builds the product library; hooks=True: ALSO the hooks build (lazily: a deployment that ships only libekfvio_hip.so imports the
    package without hipcc, and nothing but the tests and the profiling scripts asks for the other one -- ADVICE r05).  Every caller takes
    the lock, decides staleness under it and links through a temporary name that is renamed into place: the ranks of a multi-GPU launch
    all come through here at import, and none of them can map a half-written library."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(HOOKS_OBJ_DIR, exist_ok=True)
    import fcntl
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        todo = [v for v in ([_Variant(False)] + ([_Variant(True)] if hooks else [])) if force or v.stale()]
        if todo:
            _build_locked(todo, force, verbose)
        return HOOKS_LIB_PATH if hooks else LIB_PATH


def _build_locked(variants, force, verbose):
    jobs, isa_jobs = [], []
    for v in variants:
        newest_hdr = max(os.path.getmtime(h) for h in v.deps() if not h.endswith(".hip"))
        vforce = force or not os.path.exists(v.flags_stamp) or open(v.flags_stamp).read() != _flags_key()  # objects built with other flags
        v.objs = []
        chol_rebuilt = False
        for src in sources():
            name = os.path.basename(src)
            obj = os.path.join(v.obj_dir, name + ".o")
            v.objs.append(obj)
            if (not vforce and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src)
                    and os.path.getmtime(obj) >= newest_hdr):
                continue
            cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(name, []) + _extra_flags() + v.defines + ["-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
            chol_rebuilt = chol_rebuilt or name == "chol.hip"
        # chol.hip carries a hand-placed, COUNTED wait (chol_persist.inc: ready[k] goes up behind `s_waitcnt vmcnt(N)`): the ISA of
        # the very compile that is being linked is checked, with the same flags and defines (the hooks build runs the abort and
        # fault-injection tests and the stamp scripts on ITS chol_persist_kernel), and the build fails on a mismatch
        if chol_rebuilt or not os.path.exists(v.isa_stamp):
            src = os.path.join(CSRC, "chol.hip")
            isa = os.path.join(v.obj_dir, "chol.device.s")
            cmd = [HIPCC] + FLAGS + FILE_FLAGS.get("chol.hip", []) + _extra_flags() + v.defines + ["-S", "--cuda-device-only", "-o", isa, src]
            if verbose:
                print(" ".join(cmd))
            isa_jobs.append((v, cmd, subprocess.Popen(cmd), isa))
    for cmd, pr in jobs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    for v, cmd, pr, isa in isa_jobs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
        if os.path.exists(v.isa_stamp):
            os.remove(v.isa_stamp)
        problems = check_counted_waits(open(isa).read())
        if problems:
            raise RuntimeError("chol.hip%s: hand-placed counted wait does not match the compiled ISA: " % (" (hooks build)" if v.hooks else "") + "; ".join(problems))
        with open(v.isa_stamp, "w") as fh:
            fh.write("ok\n")
    for v in variants:
        tmp = "%s.tmp.%d" % (v.lib, os.getpid())
        cmd = [HIPCC, "--offload-arch=gfx950", "-fPIC", "-shared", "-Wl,--version-script=" + VERSION_SCRIPT, "-o", tmp] + v.objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp, v.lib)  # atomic: a concurrent loader maps either the old or the new complete file
        with open(v.flags_stamp + ".tmp", "w") as fh:
            fh.write(_flags_key())
        os.replace(v.flags_stamp + ".tmp", v.flags_stamp)
    return LIB_PATH


def check_counted_waits(isa_text):
    """Every inline-asm `s_waitcnt vmcnt(N)`, N > 0, of chol_persist_kernel must have exactly N vector-memory LOADS, and
    nothing else that vmcnt counts, in the straight-line code in front of it (vmcnt retires loads and stores in order: the
    wait then means "everything older than these N loads -- the factor's stores -- has completed").  Returns a list of
    problems (empty = sound)."""
    import re
    m = re.search(r"^_ZN\S*chol_persist_kernel\S*:[^\n]*\n(.*?)codeLenInByte", isa_text, re.S | re.M)
    if not m:
        return ["chol_persist_kernel not found in the ISA"]
    lines = m.group(1).splitlines()
    problems = []
    waits = [i for i, l in enumerate(lines) if re.search(r"s_waitcnt vmcnt\((\d+)\)", l) and i > 0 and "ASMSTART" in lines[i - 1]]
    counted = [i for i in waits if int(re.search(r"vmcnt\((\d+)\)", lines[i]).group(1)) > 0]
    if not counted:
        # the chain's hit path raises ready[k] behind `s_waitcnt vmcnt(20)` (chol_persist.inc): if no counted inline-asm wait is found at all,
        # the asm marker format has changed or the path has been restructured -- either way the guard below would check nothing
        problems.append("no counted inline-asm s_waitcnt vmcnt(N > 0) found in chol_persist_kernel")
    for w in counted:
        want = int(re.search(r"vmcnt\((\d+)\)", lines[w]).group(1))
        # walk back over the straight-line code in front of the wait (the stores sit in front of the branch that leads here:
        # chol_persist.inc issues them at the end of the previous step, before anything of this path)
        loads = 0
        for i in range(w - 1, -1, -1):
            l = lines[i].strip()
            if re.match(r"(global|buffer)_load_", l):
                loads += 1
            elif re.match(r"(global|buffer|flat|scratch)_(store|atomic)", l) or l.startswith(".LBB") or l.startswith("s_cbranch") or l.startswith("s_branch"):
                break
        if loads != want:
            problems.append("vmcnt(%d) behind %d loads" % (want, loads))
        # ... and the factor's write-through (sc1) stores must be what lies in front of those loads in program order: the nearest
        # vector-memory store above the wait has to be an sc1 store
        for i in range(w - 1, -1, -1):
            l = lines[i].strip()
            if re.match(r"(global|buffer|flat|scratch)_store", l):
                if " sc1" not in l:
                    problems.append("the store in front of vmcnt(%d) is not a write-through (sc1) store: %s" % (want, l))
                break
        else:
            problems.append("no store in front of vmcnt(%d)" % want)
    return problems


HOST_DIR = os.path.join(_HERE, "host")
REPLAY_PATH = os.path.join(LIB_DIR, "ekfvio_replay")


def build_host(force=False, verbose=False):
    """C++ host shim + ROS-free replay driver (g++, links the C-ABI library only)."""
    src = os.path.join(HOST_DIR, "replay_main.cpp")
    deps = [src, os.path.join(HOST_DIR, "ekfvio.hpp"), LIB_PATH]
    if (not force and os.path.exists(REPLAY_PATH)
            and all(os.path.getmtime(REPLAY_PATH) >= os.path.getmtime(d) for d in deps)):
        return REPLAY_PATH
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-o", REPLAY_PATH, src, "-L" + LIB_DIR, "-lekfvio_hip",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + LIB_DIR, "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return REPLAY_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True, hooks=True))
    print(build_host(force=True, verbose=True))
