"""In-tree build of the native libraries (hipcc cross-compiles gfx950 without a GPU).

    python -m ellp_amd.build          # builds ellp_amd/libellp_hip.so (+ libellp_host.so)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
INCLUDE = os.path.join(ROOT, "include")

ENGINE_SRC = [os.path.join(HERE, "csrc", "engine", "ellp_engine.hip"),
              os.path.join(HERE, "csrc", "engine", "ellp_qr.hip"),
              os.path.join(HERE, "csrc", "engine", "ellp_lu.hip")]
ENGINE_LIB = os.path.join(HERE, "libellp_hip.so")
HOST_DIR = os.path.join(HERE, "csrc", "host")
HOST_LIB = os.path.join(HERE, "libellp_host.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIPFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
            "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-unused-function", "-I" + INCLUDE,
            "-I" + os.path.join(HERE, "csrc", "engine")]


def engine_source_hash():
    """sha1 (12 hex digits) over the engine's sources and headers: profiles/ files carry it, so that bench.py can tell a
    counter / kernel-time file measured on THESE kernels from one measured on an older build"""
    import hashlib
    h = hashlib.sha1()
    edir = os.path.join(HERE, "csrc", "engine")
    files = sorted(os.path.join(edir, f) for f in os.listdir(edir) if f.endswith((".hip", ".inc")))
    files += sorted(os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _deps(extra):
    hdrs = [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    edir = os.path.join(HERE, "csrc", "engine")
    incs = [os.path.join(edir, f) for f in os.listdir(edir) if f.endswith(".inc")]
    return list(extra) + hdrs + incs


def build_engine(force=False, verbose=False):
    if force or _stale(ENGINE_LIB, _deps(ENGINE_SRC)):
        cmd = [HIPCC] + HIPFLAGS + ENGINE_SRC + ["-o", ENGINE_LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return ENGINE_LIB


ENGINE_DBG_LIB = os.path.join(HERE, "libellp_hip_dbg.so")


def build_engine_debug(force=False, verbose=False):
    """the same engine with -DELLP_DEBUG_BOUNDS (index assertions at every committed decision); tests load it through
    ELLP_HIP_LIB in a child process"""
    if force or _stale(ENGINE_DBG_LIB, _deps(ENGINE_SRC)):
        cmd = [HIPCC] + HIPFLAGS + ["-DELLP_DEBUG_BOUNDS"] + ENGINE_SRC + ["-o", ENGINE_DBG_LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return ENGINE_DBG_LIB


def build_host(force=False, verbose=False):
    if not os.path.isdir(HOST_DIR):
        return None
    srcs = sorted(os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR) if f.endswith(".cpp"))
    hdrs = sorted(os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR) if f.endswith(".h"))
    if not srcs:
        return None
    if force or _stale(HOST_LIB, _deps(srcs + hdrs + [ENGINE_LIB])):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-Wextra",
               "-I" + INCLUDE, "-I" + HOST_DIR] + srcs + ["-o", HOST_LIB,
               "-L" + HERE, "-lellp_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HOST_LIB


def build_all(force=False, verbose=False, debug_bounds=False):
    build_engine(force, verbose)
    build_host(force, verbose)
    if debug_bounds:
        build_engine_debug(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True, debug_bounds="--debug-bounds" in sys.argv)
